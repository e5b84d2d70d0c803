"""docs/s of sjmi_parse_batch_device on ONE GPU at 125 k / 250 k / 500 k / 1 M documents of the configs[3] set (what a rank of a
strong-scaled 8 / 4 / 2 / 1-GPU run holds): predicts the strong-scaling curve and shows where fixed per-step costs (launches,
memsets) start to matter.  Run on the GPU box; writes gpurun_out/batch_size_sweep.json (copy to profiles/rN/)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="125000,250000,500000,1000000")
    ap.add_argument("--batch-steps", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "batch_size_sweep.json"))
    a = ap.parse_args()
    import torch
    import simdjson_java_amd as S
    import workloads as W
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    work = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work)
    args = argparse.Namespace(docs=0, batch_steps=a.batch_steps, sample=1000)
    rows = []
    for n in [int(x) for x in a.sizes.split(",")]:
        r = bench.batch_single_gpu(torch, S, W, dev, work, args, with_h2d=False, n_docs=n, check=(n <= 125000), rejected=False)
        rows.append({"documents": n, "docs_per_s": r["value"], "ms_per_batch": r["ms_per_batch"], "bytes": r["counts"]["structurals"] and r["roofline"]["algorithmic_bytes_per_launch"],
                     "roofline_frac": r["roofline"]["frac"]})
        print(rows[-1], flush=True)
    full = rows[-1]["docs_per_s"]
    for r in rows:
        r["relative_to_largest"] = round(r["docs_per_s"] / full, 4)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({"what": "sjmi_parse_batch_device, device-resident, one GPU, unique ~1 KB documents (tools/docgen.c)", "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
