#!/usr/bin/env python3
"""Turn the PMC summaries of one GPU call into profiles/<kernel>_traffic.json, stamped with the SHA-256 of the kernel
sources the measurement ran on (bench.py refuses a traffic file whose stamps no longer match the tree).

  python tools/make_traffic_json.py <tag> [--commit <sha>]     (reads gpurun_out/r03_<tag>_pmc_<kernel>_{FETCH,WRITE}_SIZE.txt
                                                                 and gpurun_out/r03_<tag>_source_sha.json, both written on the
                                                                 GPU box by tools/run_final_profile_r03.sh)

HBM bytes per launch = 2 x FETCH_SIZE (gfx950: 128-B read requests are tallied at 64 B, MI355X_MICROARCH.md "HBM";
validated in round 1 on gram_group_reduce_kernel's known bytes) + WRITE_SIZE, in KB as rocprofv3 reports them, summed over
the kernels of the launch group, mean per dispatch."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {   # tag in the PMC file names -> (output file, kernels of one launch, sources, algorithmic bytes, description)
    "fwd": ("fwd_traffic.json", ("fwd_batch_kernel",), ("pleas_merging_amd/csrc/conv_fwd.hip", "pleas_merging_amd/csrc/common.hpp"),
            4237100000.0, "one ResNet-101 PLeaS update: 105 merged layers, batch 16, full merge (tools/hipbench/fwd_batch_rn101.hip on lists/rn101_dense_downsample.txt: the three strided 1x1 layers as the dense layers the subsampled merge makes of them)"),
    "wgrad": ("wgrad_traffic.json", ("wgrad_batch_kernel",), ("pleas_merging_amd/csrc/conv.hip", "pleas_merging_amd/csrc/common.hpp"),
              2159300000.0, "one ResNet-101 PLeaS update: all 105 merged layers, batch 16 (wgrad_batch_rn101.hip on lists/rn101_dense_downsample.txt; the 3-channel stem as virtual-channel rows since round 4)"),
    "gram": ("gram_traffic.json", ("gram_batch_kernel",), ("pleas_merging_amd/csrc/gram.hip", "pleas_merging_amd/csrc/common.hpp"),
             5132100000.0, "one ResNet-101 matching batch as bench.py runs it: 240 nodes contracted, 104 BatchNorm nodes derived (gram_batch_rn101.hip, rn101_nodes_derived.txt)"),
    "neq": ("neq_traffic.json", ("neq_batch_kernel",), ("pleas_merging_amd/csrc/normal_eq.hip", "pleas_merging_amd/csrc/common.hpp"),
            1936100000.0, "one ResNet-101 closed-form batch: A += U^T U of 104 layers, batch 16 (neq_batch_rn101.hip); algorithmic bytes = the "
            "merged inputs once (1.000 GB) + one read-modify-write of every A's lower triangle (sum K (K + 1) / 2 x 8 B = 0.936 GB)"),
}
LINE = re.compile(r"^(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+mean=(\S+)\s+sum=(\S+)")


def read(path, kernels):
    total, per = 0.0, {}
    for ln in open(path):
        m = LINE.match(ln)
        if m and any(k in m.group(1) for k in kernels):
            per[m.group(1).strip()] = float(m.group(4))
            total += float(m.group(4))
    return total, per


def main():
    tag = sys.argv[1]
    commit = sys.argv[sys.argv.index("--commit") + 1] if "--commit" in sys.argv else None
    rnd = sys.argv[sys.argv.index("--round") + 1] if "--round" in sys.argv else "r03"      # file prefix of the GPU call's outputs
    out_dir = os.path.join(ROOT, "gpurun_out")
    shas = json.load(open(os.path.join(out_dir, "%s_%s_source_sha.json" % (rnd, tag))))
    for key, (fname, kernels, sources, algo, what) in KERNELS.items():
        fetch_p = os.path.join(out_dir, "%s_%s_pmc_%s_FETCH_SIZE.txt" % (rnd, tag, key))
        write_p = os.path.join(out_dir, "%s_%s_pmc_%s_WRITE_SIZE.txt" % (rnd, tag, key))
        if not (os.path.exists(fetch_p) and os.path.exists(write_p)):
            print("skip", key, "(no PMC summary for tag %s)" % tag)
            continue
        fetch, per_f = read(fetch_p, kernels)
        write, per_w = read(write_p, kernels)
        if fetch <= 0:
            print("skip", key, "(kernel not found in the summary)")
            continue
        blob = {"kernel": " + ".join("pleas::" + k for k in kernels), "round": int(rnd[1:]), "tag": tag, "commit": commit, "what": what,
                "command": "tools/" + rnd + "/final_profile.sh" + " %s (standalone replay; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                           "separate passes, mean per dispatch, summed over the launch's kernels)" % tag,
                "fetch_size_kb_raw": fetch, "write_size_kb_raw": write, "fetch_size_kb_raw_per_kernel": per_f,
                "write_size_kb_raw_per_kernel": per_w,
                "correction": "gfx950: FETCH_SIZE tallies 128-B read requests at 64 B -> x2; WRITE_SIZE exact",
                "hbm_bytes_per_launch": round((2 * fetch + write) * 1024), "algorithmic_bytes_per_launch": algo,
                "source_sha256": {src: shas[src] for src in sources}}
        with open(os.path.join(ROOT, "profiles", fname), "w") as f:
            json.dump(blob, f, indent=1)
        print(fname, "hbm bytes per launch %.3e" % blob["hbm_bytes_per_launch"], "(algorithmic %s)" % algo)


if __name__ == "__main__":
    main()
