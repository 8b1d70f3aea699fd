"""Edit distance, schedule 13: when did every band finish its groups 1 / 100 / 300 / 500?  The distance between neighbouring bands
in time, along the pipeline and along the run (libbmx_exp.so, bmx_exp_ed_stamps).  python tools/ed_band_clock.py [n] [lag]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(5)
x = torch.from_numpy((rng.integers(0, 4, n) + 65).astype(np.uint8)).cuda()
z = torch.from_numpy((rng.integers(0, 4, n) + 65).astype(np.uint8)).cuda()
ctx = host.Context(0, library=host.exp_lib())
ctx.set_ed_variant(13)
if len(sys.argv) > 2:
    ctx.set_knob("ed_lag", int(sys.argv[2]))
for _ in range(3):
    d = ctx.edit_distance_device(x, z)
st = ctx.ed_stamps()
clk = np.array(st["band_clock"], dtype=np.float64) / 100.0  # us
bands = (n + 2047) // 2048
fwd = clk[:bands]
for k, g in enumerate((1, 100, 300, 500)):
    col = fwd[:, k]
    ok = col > 0
    gaps = np.diff(col)[ok[1:] & ok[:-1]]
    print(json.dumps({"group": g, "bands_that_have_it": int(ok.sum()), "us_behind_the_band_in_front_by_band": [round(float(v), 2) for v in gaps],
                      "mean_us": round(float(gaps.mean()), 2) if gaps.size else None}))
ctx.set_knob("ed_stamp_block", 5)
ctx.edit_distance_device(x, z)
import ctypes as C
out = (C.c_uint64 * (24 + 64 * 4))()
ctx._L.bmx_exp_ed_stamps(ctx._h, out)
t = [int(v) for v in out]
p_g0_end, c_g0_end = t[24 + 4 * 5], t[24 + 4 * 6]
print(json.dumps({"start_of_bands_5_and_6_us_after_band_5_finished_its_group_0": {
    "band_5_finished_group_2": (t[10] - p_g0_end) / 100.0, "band_6_starts_group_0": (t[11] - p_g0_end) / 100.0,
    "band_6_finished_group_0": (c_g0_end - p_g0_end) / 100.0}}))
print(json.dumps({"ms": ctx.last_edit_distance_ms(), "distance": d}))
