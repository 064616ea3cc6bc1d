"""DESIGN.md quotes its measured figures in generated blocks (tools/design_table.py) -- this test regenerates them from the
committed records under profiles/ and fails when the document says something else.  (Round 3's DESIGN.md quoted 10.2 us for a
row whose record said 20.9: the numbers had been typed by hand.)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_md_matches_the_committed_records():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_table.py"), "--check"], capture_output=True, text=True)
    assert p.returncode == 0, "DESIGN.md differs from the records (python tools/design_table.py --write):\n" + p.stdout[-4000:] + p.stderr[-2000:]


def test_every_generated_block_is_present_and_filled():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import design_table
    for name in design_table.BLOCKS:
        m = re.search(r"<!-- GENERATED:%s BEGIN[^\n]*-->\n(.*?)<!-- GENERATED:%s END -->" % (name, name), text, re.S)
        assert m, name
        assert len(m.group(1).strip().splitlines()) >= 2, name


def test_the_records_the_tables_come_from_are_the_bench_s_own_command():
    import json
    side = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_extra.json")))
    assert side["line_of"].startswith("bench.py --gpus 1 --steps 20 --warmup 5")
    line = side["line"]
    assert line["config"]["name"] == "cfg4_1gpu" and line["dtype"] == "f64" and line["n_gpus"] == 1
    assert line["roofline"]["kernel"].startswith("kf_step_population_kernel<double>") and 0.4 <= line["roofline"]["frac"] <= 1.0
    names = {e["name"] for e in side["extra"]}
    assert {"cfg2", "cfg3", "cfg4", "cfg4_64", "cfg5", "cfg4_64_live", "ar4m64", "uv10m"} <= names
    assert not [e for e in side["extra"] if "error" in e]
