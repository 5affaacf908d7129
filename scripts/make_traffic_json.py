#!/usr/bin/env python3
"""profiles/pmc_traffic.json from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/prof_run.py (separate runs).
HBM bytes per launch = 2 * FETCH_SIZE (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section)
+ WRITE_SIZE, both in KiB.

Only the LAST `steps` dispatches of every kernel name are averaged: the load-time calibration launches the same kernels on
16-CU sub-batches before the timed batches, and averaging over all dispatches of a name diluted the figures of every kernel
that also runs there (round 2: stem_block / block32 / 32->64 / chain<64> were reported at 2/3 of their real traffic).  The
dispatch count that went into each average is recorded and the averaged dispatches must agree within 5 %.

usage: make_traffic_json.py <fetch_csv> <write_csv> <batch> <out.json> <tag> <source_sig> [steps = 2]"""
import collections
import csv
import json
import re
import sys

H_BY_COUT = {32: 64, 64: 32, 128: 16, 256: 8}


def per_kernel(path, counter, steps):
    acc = collections.defaultdict(list)
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r.get("Dispatch_Id", 0) or 0))
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        m = re.search(r"conv_(?:mfma|ring_dma)_kernel<(\d+), (\d+), (\d)", k)
        if m:
            cin, cout, st = map(int, m.groups())
            name = f"conv3x3_s{st}_{cin}to{cout}_h{H_BY_COUT[cout]}" + ("+sc" if st == 2 else "")
        elif (mc := re.search(r"chain_kernel<([^>]*)>", k)):
            # <C, HL, SPW_L, WCB, WPB, WAVES_C, WAVES_P, GT, RB, FD, MINW, NCONV, SPLIT_ROLES, S2, KEEP>
            targs = [x.strip() for x in mc.group(1).split(",")]
            c, hl, nconv = int(targs[0]), int(targs[1]), int(targs[11])
            if len(targs) > 13 and targs[13] == "true":
                name = f"stage_{c}_h{1 << hl}(s2+sc,conv2,conv1,conv2)"
            else:
                name = f"chain{nconv}_s1_{c}_h{1 << hl}(conv2+conv1+conv2)"
        elif "layer0_stream_kernel" in k:  # (the names run_layer0_stream profiles under -- mlt_profile_entry.name holds 47 characters: with / without layer1's stride-2 conv as a fifth stage)
            # (template arguments since round 6: <F5, M16> -- the first one says whether the fifth stage rides along)
            name = "layer0_stream_h64(stem+layer0+layer1.0.conv1+sc)"[:47] if re.search(r"layer0_stream_kernel<true", k) else "layer0_stream_h64(stem+layer0.0+layer0.1)"
        elif "layer1_stream_kernel" in k:
            name = "layer1_stream_h32(conv2+conv1+conv2)"
        elif "stem_block_kernel" in k:
            name = "stem+block_s2_2to32_h64(layer0.0)"
        elif "block32_kernel" in k:
            name = "block_s1_32_h64(conv1+conv2)"
        elif "stem5_kernel" in k:
            name = "stem5x5_s2_2to32_h64+sc"
        elif "heads_kernel" in k:
            name = "heads"
        else:
            continue
        acc[name].append(float(r["Counter_Value"]))
    out = {}
    for k, v in acc.items():
        last = v[-steps:]  # the batch launches come last (calibration / warm-up dispatches of the same kernel precede them)
        mean = sum(last) / len(last)
        spread = (max(last) - min(last)) / mean if mean > 0 else 0.0
        out[k] = {"mean": mean, "n": len(last), "of": len(v), "spread": spread}
    return out


def main():
    fetch, write, batch, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    tag, sig = (sys.argv[5], sys.argv[6]) if len(sys.argv) > 6 else ("?", "?")
    steps = int(sys.argv[7]) if len(sys.argv) > 7 else 2
    f, w = per_kernel(fetch, "FETCH_SIZE", steps), per_kernel(write, "WRITE_SIZE", steps)
    # _meta.source_sig: bench.py only quotes these figures while the kernel / runtime sources still hash to it
    res = {"_meta": {"tag": tag, "source_sig": sig, "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 scripts/prof_run.py 4096 2"}}
    for name in f:
        wv = w.get(name, {"mean": 0.0, "n": 0, "of": 0, "spread": 0.0})
        assert f[name]["n"] == steps or f[name]["of"] < steps, (name, f[name])
        res[f"{name}@{batch}"] = {"batch": batch, "fetch_kib_raw": f[name]["mean"], "write_kib": wv["mean"],
                                  "hbm_bytes_per_launch": (2.0 * f[name]["mean"] + wv["mean"]) * 1024.0,
                                  "dispatches_averaged": f[name]["n"], "dispatches_of_this_name": f[name]["of"],
                                  "spread_fetch": round(f[name]["spread"], 4), "spread_write": round(wv["spread"], 4),
                                  "note": "2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts wide reads at half); last `dispatches_averaged` dispatches only"}
        if f[name]["n"] == steps and max(f[name]["spread"], wv["spread"]) > 0.05:
            res[f"{name}@{batch}"]["warning"] = "averaged dispatches differ by more than 5 %: not all of them are batch launches"
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
