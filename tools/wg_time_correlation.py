"""Are the slow workgroups of one launch the slow ones of the next?  Per-workgroup loop times of the clock-stamp build
(variant 37), several launches in one process: correlation between launches and between processes (run twice).
    python tools/wg_time_correlation.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host
host.use_library(os.environ.get("BMX_LIB", "exp"))  # every slot of the kernel table: libbmx_exp.so (BMX_LIB=<path>: another build, A/B runs)

spec = corpus.CorpusSpec("stamps", 4 << 30, 16, kind=0, seed=0x5EED0002)
ctx = host.Context(0)
d = spec.device_text(ctx)
out = torch.empty(1 << 16, dtype=torch.int64, device="cuda")
ctx.set_variant(37)
runs = []
for i in range(6):
    ctx.enqueue(d, spec.pattern(), out)
    ctx.finish(out)
    torch.cuda.synchronize()
    st = ctx.scan_stamps().astype(np.float64)
    nw = ctx.geometry(16)["block"] // 64
    rt = st[:, 6].reshape(-1, nw).max(axis=1) / 100.0
    runs.append(rt)
R = np.array(runs[1:])
print("workgroups", R.shape[1], "mean us per launch", np.round(R.mean(axis=1), 1).tolist())
print("max / median per launch", np.round(R.max(axis=1) / np.median(R, axis=1), 4).tolist())
c = np.corrcoef(R)
print("correlation between launches:\n", np.round(c, 2))
m = R.mean(axis=0)
print("per-workgroup mean over launches: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (m.min(), np.percentile(m, 10), np.median(m), np.percentile(m, 90), m.max()))
print("if every workgroup got tiles in proportion to its mean speed: slowest / median would be", round(float((R / m).max(axis=1).mean() / np.median(R / m)), 4))
np.save("gpurun_out/wg_times.npy", R)
