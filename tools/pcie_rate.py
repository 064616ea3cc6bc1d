#!/usr/bin/env python3
"""PCIe-inclusive tick rates (never bench.py's `value`): measurements start in HOST memory every tick.
  (a) target_manager_update_meas_batch: ids + AoS doubles in pageable host arrays (id lookup, H2D, pack, indexed step, sync)
  (b) dense path: SoA measurements in pinned host memory, async H2D into a device buffer, then target_batch_step
usage: python tools/pcie_rate.py [--targets 10000,1000000] [--model uniform_velocity] [--dtype f64]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", default="10000,1000000")
    ap.add_argument("--model", default="uniform_velocity")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--ticks", type=int, default=30)
    args = ap.parse_args()
    import torch
    import target_estimation_amd as te
    path = os.path.join(ROOT, "models", "model_%s_params.yaml" % args.model)
    for n in [int(x) for x in args.targets.split(",")]:
        rng = np.random.default_rng(1)
        ids = np.arange(n, dtype=np.uint32)
        p0 = np.concatenate([rng.uniform(-1, 1, (n, 3)), np.tile([0, 0, 0, 1.0], (n, 1))], 1)
        mgr = te.TargetManager(path, dtype=args.dtype)
        mgr.init_batch(ids, 0.004, 0.0, p0)
        b = mgr.batches()[0]
        meas = p0.copy()
        mgr.update_batch(ids, 0.004, meas)
        t0 = time.perf_counter()
        for _ in range(args.ticks):
            mgr.update_batch(ids, 0.004, meas)
        ta = (time.perf_counter() - t0) / args.ticks
        # the same call with the ids in random order (no slot-order fast path: one id lookup per row) and the getters
        perm = rng.permutation(n)
        ids_r, meas_r = ids[perm], np.ascontiguousarray(meas[perm])
        mgr.update_batch(ids_r, 0.004, meas_r)
        t0 = time.perf_counter()
        for _ in range(max(3, args.ticks // 5)):
            mgr.update_batch(ids_r, 0.004, meas_r)
        tr = (time.perf_counter() - t0) / max(3, args.ticks // 5)
        t0 = time.perf_counter()
        for _ in range(max(3, args.ticks // 5)):
            mgr.get_est_batch(ids_r)
        tg = (time.perf_counter() - t0) / max(3, args.ticks // 5)
        print("%s %s N=%d: by-ids in random order %.1f us/tick | get_est_batch (pose+twist+acc to host, random order) %.1f us" % (args.model, args.dtype, n, tr * 1e6, tg * 1e6), flush=True)
        pinned = torch.from_numpy(np.ascontiguousarray(meas.T)).to(b.torch_dtype()).contiguous().pin_memory()
        dev = torch.empty_like(pinned, device="cuda")
        dev.copy_(pinned, non_blocking=True); b.step(0.004, dev); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.ticks):
            dev.copy_(pinned, non_blocking=True)
            b.step(0.004, dev)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) / args.ticks
        t0 = time.perf_counter()
        for _ in range(args.ticks):
            b.step(0.004, dev)
        torch.cuda.synchronize()
        tc = (time.perf_counter() - t0) / args.ticks
        rows = pinned if args.model.startswith("angular") else pinned[:3].contiguous().pin_memory()
        b.step_host(0.004, rows)
        t0 = time.perf_counter()
        for _ in range(args.ticks):
            b.step_host(0.004, rows)
        th = (time.perf_counter() - t0) / args.ticks
        print("%s %s N=%d: target_batch_step_host (pinned SoA rows in the batch precision, %d rows) %.1f us/tick (%.3g cycles/s)"
              % (args.model, args.dtype, n, rows.shape[0], th * 1e6, n / th), flush=True)
        print("%s %s N=%d: by-ids host arrays %.1f us/tick (%.3g cycles/s) | pinned SoA H2D + step %.1f us/tick (%.3g cycles/s) | device-resident %.1f us/tick (%.3g cycles/s)"
              % (args.model, args.dtype, n, ta * 1e6, n / ta, tb * 1e6, n / tb, tc * 1e6, n / tc), flush=True)
        mgr.close()


if __name__ == "__main__":
    main()
