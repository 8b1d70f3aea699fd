"""Where a step of the helper-wave edit-distance band (ed variant 13, groups of 32 steps of two rows) spends its cycles: the same
kernel with parts of the step left out (libbmx_exp.so, knob ed_step_x; the distances of those builds are WRONG on purpose).
python tools/ed_step_experiments.py [n]"""
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
ctx.set_knob("ed_lag", 310)
names = ["the product step", "band edge not collected", "no Eq-word requests", "row_shr instead of wave_shr", "no feed-ring requests",
         "no hand to the right", "no edge collection, no Eq-word requests", "no LDS instruction at all", "no LDS instruction, no DPP"]
for xi, name in enumerate(names):
    ctx.set_knob("ed_step_x", xi)
    ms = []
    for _ in range(4):
        d = ctx.edit_distance_device(x, z)
        ms.append(ctx.last_edit_distance_ms())
    st = ctx.ed_stamps()
    print(json.dumps({"experiment": xi, "what": name, "distance": d, "ms": round(min(ms[1:]), 3),
                      "cycles_per_step": round(st["cycles_in_steps"] / max(1, st["groups"] * st["steps_per_group"]), 1),
                      "cycles_between_per_group": round(st["cycles_between"] / max(1, st["groups"]), 1),
                      "of_which_waiting": round(st["cycles_validate"] / max(1, st["groups"]), 1)}), flush=True)
