#!/usr/bin/env python3
"""Build-time tuning sweep: `build` (CPU box) compiles variant libraries under gpurun_out-independent
`fastintercu-vvc_amd/_variants/`; `run` (GPU box) benches each one and prints per-kernel ms."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "fastintercu-vvc_amd", "_variants")
VARIANTS = {  # name -> -D defines (the CFG_* knobs in csrc/mlt_kernels.hip); run each twice for box noise
    "base": [],
    # round 4: taps per weight step of the exact arithmetic's per-conv kernels (SWEEP_ARGS="--weight-seed 22" = the exact 128 model, "--size 64")
    # (gte_s2_2 / gte_s2_5 / gte_s1_3: profiles/r04h_sweep_exact*.txt -- CFG_GTE_S2 = 2 became the default)
    # patch items per lane prefetched in registers, exact arithmetic (phase stamps: the synchronous tail of the commit is 36 - 45 % of the 32-channel launches)
    # (une_5 / une_5_all / une_6: profiles/r04m_sweep_exact_prefetch.txt -- CFG_UNE_32 = 5 became the default)
    # 32 -> 32 per-conv kernels on 256-pixel tiles (4 waves): the exact form then fits two workgroups per CU (78 KiB of LDS each instead of 122)
    # (x32_wp4 / x32_wp4_une3 of round 4: CFG_32_WP=4 [, CFG_UNE_32=3])
    # round 5: layer0_stream_kernel -- SIMD mapping, fragment depth, packed epilogues, knock-outs (1 no fragment reads, 2 no MFMAs, 4 no epilogues, 8 no barrier)
    # (first form, profiles/r05o_l0_sweep1.txt: map1 0.929, pd6 0.864, pkepi 0.847, no fragment reads 0.876, no MFMAs 0.632, no epilogues 0.639, no barrier 0.859 against 0.870)
    # (profiles/r05o_l0_sweep2.txt: wave priorities by stage 0.910 / 0.908 / 0.887, no cross-barrier prefetch 0.879 against 0.894 -- all noise)
    # (a second accumulator for the odd items, to see whether the dependent MFMA chain limits a wave, spilled 24 VGPRs at 128; scripts/probes/mfma_chain_probe.hip
    #  answers the question in isolation: it does not -- one wave with one chain already runs the pipe at the register-only rate)
    # (the five-stage form's knobs -- cross-barrier prefetch, raw-row group -- all within noise: profiles/r05o_l0_sweeps.txt)
    # layer1_stream_kernel: fragment depth, knock-outs (1 no epilogue waves, 2 no accumulator hand-over, 4 loaders idle)
    "l1_pd6": ["CFG_L1_PD=6"],
    "l1_ko1": ["CFG_L1_KO=1"],
    "l1_ko2": ["CFG_L1_KO=2"],
    "l1_ko4": ["CFG_L1_KO=4"],
}
# round 3, 32->64 stride-2 kernel (0.446 ms): all slower -- 256-pixel tiles on 16 waves 0.81, 64 couts per wave 0.94, both 0.54, UN 6 0.46
#   "s2_wp8": ["CFG_3264_WP=8"], "s2_wcb2_wp8": ["CFG_3264_WCB=2", "CFG_3264_WC=1", "CFG_3264_WP=8"]
# (the KO_* knock-out knobs of rounds 1-3 were removed from the kernels at the end of round 3; their results are in DESIGN.md)


def main():
    import mltcnn_pkg
    pkg = mltcnn_pkg.load()
    if sys.argv[1] == "build":
        os.makedirs(VDIR, exist_ok=True)
        from concurrent.futures import ThreadPoolExecutor
        def one(item):
            name, defs = item
            pkg.build.build_lib(defines=defs, out=os.path.join(VDIR, f"lib_{name}.so"))
            print("built", name, flush=True)
        with ThreadPoolExecutor(max_workers=int(os.environ.get("SWEEP_JOBS", "4"))) as ex:
            list(ex.map(one, VARIANTS.items()))
    else:
        for name in VARIANTS:
            env = dict(os.environ, MLT_TUNING="1", MLT_LIB_PATH=os.path.join(VDIR, f"lib_{name}.so"), MLT_CHUNK="4096")
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "10", "--no-cpu-baseline", "--flags", os.environ.get("SWEEP_FLAGS", "0")] + os.environ.get("SWEEP_ARGS", "").split(),
                                 env=env, capture_output=True, text=True).stdout
            line = [l for l in out.splitlines() if l.startswith("{")]
            if not line:
                print(name, "FAILED")
                continue
            d = json.loads(line[-1])
            print(f"{name:16s} {d['value']:10.0f} CU/s  err {d['parity']['max_abs_dlogit']:.1e}  " +
                  " ".join(f"{k['avg_ms']:.3f}" for k in d["derived"]["kernels"]))
            if os.environ.get("SWEEP_NAMES") and name == list(VARIANTS)[0]:
                print(" " * 44 + " ".join(k["name"][:5] for k in d["derived"]["kernels"]))


if __name__ == "__main__":
    main()
