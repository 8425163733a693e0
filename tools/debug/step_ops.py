"""Which Python lines launch the torch-native (non-library) kernels of the timed step: torch.profiler with stacks over
three eager steps, device time aggregated per (aten op, innermost gcanet_amd/bench frame)."""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from gcanet_amd import dgcnn  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=64, dtype="bf16").to(dev)
    pts, nrm = bench.synth_clouds(range(8), 8192, dev)
    st = bench.make_step(model, pts, nrm, 1)
    for _ in range(3):
        st["step"]()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    steps = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        for _ in range(steps):
            st["step"]()
        torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        dt = getattr(ev, "self_device_time_total", 0) or 0
        if dt <= 0 or not ev.name.startswith("aten::"):
            continue
        where = "?"
        for fr in ev.stack or []:
            if "gcanet_amd/" in fr or "bench.py" in fr:
                where = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr[-70:]
                break
        where = where + "  " + str([tuple(x) if isinstance(x, (list, tuple)) else x for x in (ev.input_shapes or [])])[:110]
        a = agg[(ev.name, where)]
        a[0] += 1
        a[1] += dt
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print("aten device time per step: %.3f ms in %.0f launches" % (tot / 1e3 / steps, sum(v[0] for _, v in rows) / steps))
    for (name, where), (c, t) in rows[:120]:
        print("%7.1f us/step %5.1f calls/step  %-28s %s" % (t / steps, c / steps, name, where))


if __name__ == "__main__":
    main()
