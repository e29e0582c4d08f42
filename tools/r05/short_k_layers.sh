#!/bin/bash
# The short-K 1 x 1 layers of a ResNet-101 one geometry at a time (8 copies of the layer per grouped launch, batch 16) on the
# forward / weight-gradient replays: time, flop rate and ALGORITHMIC bytes per second of each -- how far is each from its own
# roofline (HBM 8 TB/s, fp32 matrix 157.3 TFLOP/s)?   Output: gpurun_out/r05_short_k_layers.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; CS=$R/pleas_merging_amd/csrc
cd $R/tools/hipbench
for k in wgrad fwd; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/${k}_replay ${k}_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || exit 1
done
{ for geo in "256 64 56 56" "64 256 56 56" "64 64 56 56" "128 256 56 56" "512 128 28 28" "128 512 28 28" "256 512 28 28" "1024 256 14 14" "256 1024 14 14"; do
    f=/tmp/short_k_list.txt; echo 8 > $f; for i in 1 2 3 4 5 6 7 8; do echo "$geo 1 1 0" >> $f; done
    for arith in fp32 split_bf16; do
      echo -n "fwd   [Cout Cin H W = $geo] x 8 $arith: "; PLEAS_ARITH=$arith timeout -k 10 60 /tmp/fwd_replay $f 30 || exit 1
      echo -n "wgrad [Cout Cin H W = $geo] x 8 $arith: "; PLEAS_ARITH=$arith timeout -k 10 60 /tmp/wgrad_replay $f 20 || exit 1
    done
  done; } > $O/r05_short_k_layers.txt 2>&1
python3 - <<'PY'
import re
out = []
for line in open("/root/repo/gpurun_out/r05_short_k_layers.txt"):
    m = re.search(r"algorithmic ([\d.]+) GFLOP ([\d.]+) MB per update; ([\d.]+) ms", line)
    if not m:
        out.append(line.rstrip()); continue
    gf, mb, ms = map(float, m.groups())
    tf, tbs = gf / ms, mb / ms / 1e3
    roof_ms = max(gf / 157.3, mb / 8000.0 / 1.0)       # ms at the fp32 matrix peak / at 8 TB/s on the algorithmic bytes
    out.append("%s  -> %.1f TF/s, %.2f TB/s algorithmic, %.2f of its own roofline (%s-bound)" % (
        line.split(": layers")[0], tf, tbs, roof_ms / ms, "HBM" if mb / 8000.0 > gf / 157.3 else "MFMA"))
open("/root/repo/gpurun_out/r05_short_k_layers.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
