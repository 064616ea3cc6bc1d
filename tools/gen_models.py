#!/usr/bin/env python3
"""Generate the four motion-model parameter files (models/model_<type>_params.yaml).

Restates the reference's offline MATLAB tooling so the parameter sets can be
synthesised for any frequency / noise levels without MATLAB:

  * Q = Gamma * diag(sigma_ddot)^2 * Gamma^T with Gamma = [dt^2/2 I; dt I; (I)]
    (reference: matlab/generateModel.m:9-23),
  * R = diag(sigma_m)^2 (matlab/generateModel.m:31-33),
  * P0 = diag(sigma_p)  -- NOT squared, as in the reference (matlab/generateModel.m:41),
  * file format `type / frequency / Q / R / P` as flat row-major flow sequences
    printed with %.20f (matlab/model2yaml.m:27-33, matlab/matlab2yaml.m:26-38),
  * the four example parameterisations (matlab/generateExamples.m).

Usage: python tools/gen_models.py [outdir]   (default: models/)
"""
import os
import sys
import numpy as np


def generate_model(sigma_ddot, sigma_m, sigma_p, frequency, accelerations):
    dt = 1.0 / frequency
    dim = len(sigma_ddot)
    eye = np.eye(dim)
    blocks = [0.5 * dt ** 2 * eye, dt * eye]
    if accelerations:
        blocks.append(1.0 * eye)
    gamma = np.vstack(blocks)
    sigma_a = np.diag(np.asarray(sigma_ddot, dtype=np.float64))
    q = gamma @ (sigma_a ** 2) @ gamma.T
    r = np.diag(np.asarray(sigma_m, dtype=np.float64)) ** 2
    p = np.diag(np.asarray(sigma_p, dtype=np.float64))
    assert q.shape == (len(sigma_p),) * 2
    return q, r, p


def _flow(name, mat):
    return "%s: [%s]\n" % (name, ", ".join("%.20f" % v for v in mat.reshape(-1)))


def write_yaml(path, type_name, frequency, q, r, p):
    with open(path, "w") as f:
        f.write("type: %s\n" % type_name)
        f.write("frequency: %f\n" % frequency)
        f.write(_flow("Q", q))
        f.write(_flow("R", r))
        f.write(_flow("P", p))


I3 = [1.0, 1.0, 1.0]


def _rep(v):
    return [v * e for e in I3]


EXAMPLES = {
    # type: (sigma_ddot, sigma_m, sigma_p, accelerations)
    "angular_rates": ([1e-3] * 3 + [1e-5] * 3, [0.01] * 3 + [0.1] * 3,
                      _rep(0.1) + _rep(0.01) * 5, True),
    "angular_velocities": ([1e-3] * 3 + [1e-5] * 3, [0.01] * 3 + [0.1] * 3,
                           _rep(0.1) + _rep(0.01) * 3, False),
    "uniform_acceleration": ([1e-3] * 3, [0.01] * 3,
                             [0.1] * 3 + [0.01] * 3 + [0.001] * 3, True),
    "uniform_velocity": ([1e-3] * 3, [0.01] * 3, [0.1] * 3 + [0.01] * 3, False),
}


def main(outdir="models", frequency=250.0):
    os.makedirs(outdir, exist_ok=True)
    for type_name, (sdd, sm, sp, acc) in EXAMPLES.items():
        q, r, p = generate_model(sdd, sm, sp, frequency, acc)
        write_yaml(os.path.join(outdir, "model_%s_params.yaml" % type_name),
                   type_name, frequency, q, r, p)


if __name__ == "__main__":
    main(*(sys.argv[1:2]))
