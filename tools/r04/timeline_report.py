"""Report of tools/r04/timeline.sh's records (gpurun_out/r04_timeline_{fwd,wgrad}.bin: per work item start, end [10 ns ticks],
phases, form and work): workgroups in flight and MFMA work rate over the launch, slot-time efficiency per tile form, phase
breakdown of the flat forms by K length."""
import sys
import numpy as np

CU_PEAK = 157.3e12 / 256           # flop / s of one CU (fp32 MFMA)
BIN = 50.0                         # us


def spread(st, en, weight, nb):
    out = np.zeros(nb)
    for s, e, w in zip(st, en, weight):
        d = max(e - s, 1e-3)
        for i in range(int(s // BIN), min(int(e // BIN), nb - 1) + 1):
            out[i] += w * max(0.0, min(e, (i + 1) * BIN) - max(s, i * BIN)) / d
    return out


def load(path, wgrad):
    a = np.fromfile(path, dtype=np.int64).reshape(-1, 4)
    if wgrad:
        a = a[(a[:, 2] != -1) & (a[:, 1] > 0)]
    t0 = a[:, 0].min()
    return a, (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0


def main(d="gpurun_out"):
    for name in ("wgrad", "fwd"):
        a, st, en = load("%s/r04_timeline_%s.bin" % (d, name), name == "wgrad")
        work = (a[:, 3] >> 8).astype(float) * 128 * 64 if name == "fwd" else a[:, 3].astype(float) * 64     # flop
        nb = int(en.max() // BIN) + 1
        print("%s: %d work items, launch %.0f us, item time summed %.1f ms = %.2f ms x 512 workgroup slots"
              % (name, len(a), en.max(), (en - st).sum() / 1e3, (en - st).sum() / 1e3 / 512))
        print("  workgroups in flight per %d us:" % BIN, " ".join("%d" % x for x in spread(st, en, (en - st) / BIN, nb)))
        rate = spread(st, en, work, nb) / (BIN * 1e-6) / 157.3e12
        print("  MFMA work rate (share of the fp32 matrix peak, padded tiles):", " ".join("%.2f" % x for x in rate))
        if name == "wgrad":
            eff = work / ((en - st) * 1e-6 * CU_PEAK / 2)
            print("  per-item rate relative to half a CU: 5 / 25 / 50 / 75 / 95 %%: %s" % np.quantile(eff, [.05, .25, .5, .75, .95]).round(2))
            continue
        form = a[:, 3] & 0xff
        pro, kl = (a[:, 2] & 0xffffffff) / 100.0, ((a[:, 2] >> 32) & 0xffffffff) / 100.0
        for f in np.unique(form):
            m = form == f
            slot = (en - st)[m].sum() * 1e-6
            line = "  form %d: %5d items, %5.1f GFLOP padded, slot time %5.1f ms (%4.1f %%), %.2f of half a CU, in flight %4.0f-%4.0f us" % (
                f, m.sum(), work[m].sum() / 1e9, slot * 1e3, 100 * slot / ((en - st).sum() * 1e-6), work[m].sum() / (slot * CU_PEAK / 2),
                st[m].min(), en[m].max())
            if f >= 4:
                line += "; item %.1f us = prologue %.1f + K loop %.1f + epilogue %.1f" % (
                    (en - st)[m].mean(), pro[m].mean(), (kl - pro)[m].mean(), (en - st - kl)[m].mean())
            print(line)
            tm = 64 if f >= 7 else 128
            chunks = (a[:, 3] >> 8) // tm
            for c in np.unique(chunks[m]) if f >= 4 else ():
                mm = m & (chunks == c)
                if mm.sum() > 200:
                    print("      %3d chunks: %5d items, %.1f us = %.1f + %.1f + %.1f (its MFMAs alone on a whole CU: %.1f us)" % (
                        c, mm.sum(), (en - st)[mm].mean(), pro[mm].mean(), (kl - pro)[mm].mean(), (en - st - kl)[mm].mean(),
                        c * 4096 * (tm / 128) / 2400 / 4))


if __name__ == "__main__":
    main(*sys.argv[1:])
