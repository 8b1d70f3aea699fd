"""Band pipeline: time vs the lag assumed when placing the cut rows (BMX_ED_LAG) and columns per lane."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(5)
x = torch.from_numpy((rng.integers(0, 4, n) + 65).astype(np.uint8)).cuda()
z = torch.from_numpy((rng.integers(0, 4, n) + 65).astype(np.uint8)).cuda()
ctx = host.Context(0, library=host.exp_lib())  # the lag / group switches exist in libbmx_exp.so only
sel = [tuple(int(t) for t in a.split(':')) for a in sys.argv[2:]] or [(13, 32), (13, 16), (11, 32), (9, 32)]
for v, grp in sel:
    ctx.set_ed_variant(v)
    ctx.set_knob("ed_group", grp)
    row = {}
    for lag in (100, 140, 180, 220, 260, 300, 350, 400, 500, 600, 800):
        ctx.set_knob("ed_lag", lag)
        ms = []
        for _ in range(4):
            d = ctx.edit_distance_device(x, z)
            ms.append(ctx.last_edit_distance_ms())
        row[lag] = round(min(ms[1:]), 3)
    out = {"variant": v, "group": grp, "distance": d, "ms_by_lag": row}
    if v in (11, 12, 13):  # where the middle band's cycles went, at the default lag
        ctx.set_knob("ed_lag", -1)
        ctx.edit_distance_device(x, z)
        st = ctx.ed_stamps()
        if st["groups"]:
            st["cycles_per_step"] = round(st["cycles_in_steps"] / (st["groups"] * st["steps_per_group"]), 1)
            st["cycles_between_per_group"] = round(st["cycles_between"] / st["groups"], 1)
        out["stamps_middle_band_default_lag"] = st
        if v == 13:  # the band nobody waits for but everybody waits on: forward band 0
            ctx.set_knob("ed_stamp_block", 0)
            ctx.edit_distance_device(x, z)
            st = ctx.ed_stamps()
            if st["groups"]:
                st["cycles_per_step"] = round(st["cycles_in_steps"] / (st["groups"] * st["steps_per_group"]), 1)
                st["cycles_between_per_group"] = round(st["cycles_between"] / st["groups"], 1)
            out["stamps_band_0_default_lag"] = st
            ctx.set_knob("ed_stamp_block", 5)  # (the timeline of a hand-over between forward bands 5 and 6)
            ctx.edit_distance_device(x, z)
            out["handover_band_5_to_6_us"] = ctx.ed_stamps().get("handover_us")
            ctx.set_knob("ed_stamp_block", -1)
    print(json.dumps(out), flush=True)
