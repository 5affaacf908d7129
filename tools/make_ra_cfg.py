#!/usr/bin/env python3
"""Write tests/data/vtm_ra_tools.cfg: a random-access encoder configuration WRITTEN FOR THIS REPOSITORY that switches on the tool set of
the reference's evaluation configuration (vtm-mlt-cpp/cfg/encoder_randomaccess_vtm.cfg: CTU 128 `:112`, MTT depth 3 `:119`, the coding
tools of `:123-168` -- LMCS, DMVR, BIO, CIIP, Geo, BCW, SMVD, MMVD, affine + AMVR, SbTMVP, SBT, MTS, LFNST, ISP, MIP, MRL, joint Cb-Cr,
dependent quantisation, ALF / CCALF, the GOP-based temporal filter), so that the GPU box -- where the reference tree does not exist -- can
run the encoder under the tools the reference's protocol runs (`script_128/*.sh`).  What is NOT taken from the reference: the GOP table.
The hierarchical-B structure below (GOP 16, dyadic) and its reference picture lists are DERIVED here: coding order by bisection, active
references = the nearest coded pictures on either side, inactive entries = every picture a later frame still needs (VVC keeps a picture
only while every following picture's lists name it).

usage: python tools/make_ra_cfg.py [--gop 16] [--out tests/data/vtm_ra_tools.cfg]"""
import argparse
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# VTM option name -> value; the tool set of the reference's evaluation protocol
TOOLS = [
    ("# partitioning", None),
    ("CTUSize", 128), ("LCTUFast", 1), ("DualITree", 1), ("MinQTLumaISlice", 8), ("MinQTChromaISliceInChromaSamples", 4), ("MinQTNonISlice", 8),
    ("MaxMTTHierarchyDepth", 3), ("MaxMTTHierarchyDepthISliceL", 3), ("MaxMTTHierarchyDepthISliceC", 3),
    ("# inter tools", None),
    ("MMVD", 1), ("Affine", 1), ("AffineAmvr", 1), ("SbTMVP", 1), ("MaxNumMergeCand", 6), ("IMV", 1), ("BCW", 1), ("BcwFast", 1), ("BIO", 1), ("CIIP", 1), ("Geo", 1),
    ("DMVR", 1), ("SMVD", 1), ("PROF", 1), ("AllowDisFracMMVD", 1), ("IBC", 0),
    ("# transform / quantisation / intra tools", None),
    ("MTS", 1), ("MTSIntraMaxCand", 4), ("MTSInterMaxCand", 4), ("SBT", 1), ("LFNST", 1), ("ISP", 1), ("MRL", 1), ("MIP", 1), ("LMChroma", 1), ("JointCbCr", 1), ("DepQuant", 1),
    ("TransformSkip", 1), ("TransformSkipFast", 1), ("TransformSkipLog2MaxSize", 5), ("ChromaTS", 1), ("RDOQ", 1), ("RDOQTS", 1),
    ("# in-loop", None),
    ("SAO", 1), ("ALF", 1), ("ALFStrength", 1.0), ("ALFAllowPredefinedFilters", 1), ("CCALFStrength", 1.0),
    ("LMCSEnable", 1), ("LMCSSignalType", 0), ("LMCSUpdateCtrl", 0), ("LMCSOffset", 6),
    ("# encoder speed-ups of the protocol", None),
    ("PBIntraFast", 1), ("ISPFast", 0), ("FastMrg", 1), ("AMaxBT", 1), ("FastMIP", 0), ("FastLFNST", 0), ("FastLocalDualTreeMode", 1), ("AffineAmvrEncOpt", 1), ("MmvdDisNum", 6),
    ("FEN", 1), ("FDM", 1), ("FastSearch", 1), ("SearchRange", 384), ("ASR", 1), ("MinSearchWindow", 96), ("BipredSearchRange", 4), ("HadamardME", 1),
    ("TemporalFilter", 1), ("TemporalFilterFutureReference", 1), ("TemporalFilterStrengthFrame8", 0.95), ("TemporalFilterStrengthFrame16", 1.5),
    ("# rate / misc", None),
    ("InternalBitDepth", 10), ("RateControl", 0), ("MaxDeltaQP", 0), ("DeltaQpRD", 0), ("SEIDecodedPictureHash", 1),
    ("SameCQPTablesForAllChroma", 1), ("QpInValCb", "17 22 34 42"), ("QpOutValCb", "17 23 35 39"),
]


def coding_order(gop):
    """key picture first, then bisection (dyadic hierarchical B): 16, 8, 4, 2, 1, 3, 6, 5, 7, 12, ..."""
    order, level = [gop], {gop: 0}

    def rec(lo, hi, lv):
        if hi - lo < 2:
            return
        mid = (lo + hi) // 2
        order.append(mid)
        level[mid] = lv
        rec(lo, mid, lv + 1)
        rec(mid, hi, lv + 1)
    rec(0, gop, 1)
    return order, level


def gop_table(gop):
    order, level = coding_order(gop)
    n_gops = 4
    seq = [(g * gop + p, p) for g in range(n_gops) for p in order]     # (absolute POC, POC within GOP) in coding order; POC 0 (intra) precedes
    coded_before = lambda i: {0} | {seq[j][0] for j in range(i)}
    active = []
    for i, (poc, _) in enumerate(seq):
        lv_of = lambda q: level[q % gop if q % gop else gop]
        own = level[seq[i][1]]
        have = {q for q in coded_before(i) if lv_of(q) < max(own, 1)}   # only pictures of a LOWER temporal layer (key pictures: other key pictures)
        past = sorted((q for q in have if q < poc), reverse=True)
        fut = sorted(q for q in have if q > poc)
        l0 = past[:2]
        l1 = fut[:2] if fut else past[:2]
        if len(l1) < 2:
            l1 = l1 + [q for q in past if q not in l1][:2 - len(l1)]
        active.append((l0, l1))
    rows = []
    base = gop  # describe the SECOND gop (steady state)
    for i, (poc, rel) in enumerate(seq):
        if not (base < poc <= 2 * base):
            continue
        have = coded_before(i)
        needed = set()
        for j in range(i, len(seq)):
            for q in active[j][0] + active[j][1]:
                if q in have:
                    needed.add(q)
        l0, l1 = active[i]
        extra = sorted(needed - set(l0) - set(l1), key=lambda q: abs(poc - q))
        l0_full = l0 + [q for q in extra]                       # inactive entries keep later frames' references alive
        rows.append((rel, level[rel], [poc - q for q in l0_full], len(l0), [poc - q for q in l1], len(l1)))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gop", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "data", "vtm_ra_tools.cfg"))
    a = ap.parse_args()
    rows = gop_table(a.gop)
    out = ["# Random-access configuration with the tool set of the reference's evaluation protocol; written by tools/make_ra_cfg.py (this repository).",
           f"# GOP {a.gop}, dyadic hierarchical B; reference picture lists derived by the script (nearest coded pictures on either side).",
           "Profile                       : auto",
           f"IntraPeriod                   : {2 * a.gop}",
           "DecodingRefreshType           : 1",
           f"GOPSize                       : {a.gop}",
           "IntraQPOffset                 : -3",
           "LambdaFromQpEnable            : 1",
           "#          Type POC QPoffset QPOffsetModelOff QPOffsetModelScale CbQPoffset CrQPoffset QPfactor tcOffsetDiv2 betaOffsetDiv2 CbTcOffsetDiv2 CbBetaOffsetDiv2 CrTcOffsetDiv2 CrBetaOffsetDiv2 temporal_id #ref_pics_active_L0 #ref_pics_L0 reference_pictures_L0 #ref_pics_active_L1 #ref_pics_L1 reference_pictures_L1"]
    for k, (rel, lv, l0, a0, l1, a1) in enumerate(rows):
        qpo = 0 if lv == 0 else lv + 1
        out.append(f"Frame{k + 1}: B {rel:3d} {qpo} 0.0 0.0 0 0 1.0 0 0 0 0 0 0 {lv} {a0} {len(l0)} {' '.join(map(str, l0))} {a1} {len(l1)} {' '.join(map(str, l1))}")
    out.append("QP                            : 32")
    for k, v in TOOLS:
        out.append(k if v is None else f"{k:<30}: {v}")
    with open(a.out, "w") as f:
        f.write("\n".join(out) + "\n")
    print(f"wrote {a.out}")


if __name__ == "__main__":
    main()
