"""Deterministic synthetic weights and CU patches (counter-based RNG).

Nothing here comes from the reference: the reference ships no trained weights
(`/root/reference/.MISSING_LARGE_BLOBS:7`) and no test vectors, so
parity fixtures, GPU tests and bench.py all regenerate the SAME weights / inputs from
integer seeds with this module.  The generator is a pure function of
(seed, stream-name, index) so tensors can be produced in any order, on any rank.

Distributions follow SURVEY.md §8(d):
  conv weights  Kaiming-normal, fan_out, relu gain   (mlt_ctu_or_pq_arch.py:258-260)
  BN            gamma~U(0.5,1.5) beta~N(0,0.1) mean~N(0,0.1) var~U(0.5,1.5)
                (NOT the identity init of :261-263 - identity BN would hide folding bugs)
  heads         U(+-1/sqrt(fan_in)) like nn.Linear's default
  patches       10-bit luma `org`, `pred = clip(org + noise)` as int16 ("Pel", TypeDef.h:277)
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array."""
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def raw_u64(seed: int, stream: str, n: int, offset: int = 0) -> np.ndarray:
    """n 64-bit words of stream `stream` under `seed`, starting at counter `offset`."""
    key = _mix(np.array([seed & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0] ^ np.uint64(_fnv1a64(stream))
    key = _mix(np.array([key], dtype=np.uint64))[0]
    ctr = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _mix(key + ctr * np.uint64(0xD1342543DE82EF95))


def uniform(seed: int, stream: str, n: int, offset: int = 0) -> np.ndarray:
    """float64 uniform in [0,1)."""
    return (raw_u64(seed, stream, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: str, n: int) -> np.ndarray:
    """float64 standard normal (Box-Muller on two sub-streams)."""
    u1 = uniform(seed, stream + "#a", n)
    u2 = uniform(seed, stream + "#b", n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)


def randint(seed: int, stream: str, n: int, lo: int, hi: int, offset: int = 0) -> np.ndarray:
    """int64 uniform in [lo, hi] inclusive."""
    span = np.uint64(hi - lo + 1)
    return (raw_u64(seed, stream, n, offset) % span).astype(np.int64) + lo


# --------------------------------------------------------------------------------------
# architecture tables (arithmetic spec: mlt_ctu_or_pq_arch.py:239-299, mlt_cu_or_pq_arch.py:59-128)
# --------------------------------------------------------------------------------------
ARCH_CTU = 0  # MltCnnL3ORPQv4 / GapBigMltCtuORPQ  (128x128)
ARCH_CU = 1   # MltCnnL4ORPQv4 / GapBigMltCuORPQ   (64/32/16)

STAGE_PLANES = {ARCH_CTU: (32, 64, 128, 256), ARCH_CU: (32, 64, 96, 128, 256)}
HEAD_CLASSES = {ARCH_CTU: (2, 3, 4), ARCH_CU: (2, 3, 4, 6)}
STEM_PLANES = 32


def arch_for_size(size: int) -> int:
    """EncCu.cpp:897-899 picks the model file by cuw: 128 -> CTU arch, 64/32/16 -> CU arch."""
    if size == 128:
        return ARCH_CTU
    if size in (64, 32, 16):
        return ARCH_CU
    raise ValueError(f"unsupported CU size {size}")


def state_dict_spec(arch: int) -> "OrderedDict[str, tuple]":
    """Ordered {state_dict key: shape} exactly as the reference module registers them
    (checked against the imported reference in tools/gen_golden.py)."""
    spec: "OrderedDict[str, tuple]" = OrderedDict()

    def bn(prefix, c):
        spec[prefix + ".weight"] = (c,)
        spec[prefix + ".bias"] = (c,)
        spec[prefix + ".running_mean"] = (c,)
        spec[prefix + ".running_var"] = (c,)
        spec[prefix + ".num_batches_tracked"] = ()

    spec["conv1.weight"] = (STEM_PLANES, 2, 3, 3)
    bn("bn1", STEM_PLANES)  # registered but never applied (mlt_ctu_or_pq_arch.py:247,277-278)
    planes = STAGE_PLANES[arch]
    heads = HEAD_CLASSES[arch]
    cin = STEM_PLANES
    for li, c in enumerate(planes):
        for bi in range(2):
            p = f"layer{li}.{bi}"
            spec[p + ".conv1.weight"] = (c, cin if bi == 0 else c, 3, 3)
            bn(p + ".bn1", c)
            spec[p + ".conv2.weight"] = (c, c, 3, 3)
            bn(p + ".bn2", c)
            if bi == 0:  # stride 2 => projection shortcut, every stage (arch:44-50)
                spec[p + ".shortcut.0.weight"] = (c, cin, 1, 1)
                bn(p + ".shortcut.1", c)
        cin = c
        if li >= 1:
            spec[f"branch{li}.weight"] = (heads[li - 1], c + 2)
            spec[f"branch{li}.bias"] = (heads[li - 1],)
    return spec


def make_state_dict(arch: int, weight_seed: int, head_scale: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Synthetic fp32 state_dict with non-trivial BN statistics."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in state_dict_spec(arch).items():
        n = int(np.prod(shape)) if shape else 1
        if key.endswith("num_batches_tracked"):
            sd[key] = np.array(1000, dtype=np.int64)
            continue
        if key.endswith("running_var"):
            v = 0.5 + uniform(weight_seed, key, n)
        elif key.endswith("running_mean"):
            v = 0.1 * normal(weight_seed, key, n)
        elif ".bn" in key or "shortcut.1" in key or key.startswith("bn1"):
            if key.endswith(".weight"):
                v = 0.5 + uniform(weight_seed, key, n)
            else:
                v = 0.1 * normal(weight_seed, key, n)
        elif key.startswith("branch"):
            fan_in = shape[-1] if key.endswith(".weight") else state_dict_spec(arch)[key[:-4] + "weight"][-1]
            bound = head_scale / math.sqrt(fan_in)
            v = (2.0 * uniform(weight_seed, key, n) - 1.0) * bound
        else:  # conv weight [Cout, Cin, kh, kw]: kaiming_normal_(mode='fan_out', relu)
            fan_out = shape[0] * shape[2] * shape[3]
            v = normal(weight_seed, key, n) * math.sqrt(2.0 / fan_out)
        sd[key] = v.astype(np.float32).reshape(shape)
    return sd


# --------------------------------------------------------------------------------------
# CU patches
# --------------------------------------------------------------------------------------
KIND_TEXTURE = 0     # blocky base + texture, pred = org + noise[-40,40]   (bench workload)
KIND_UNIFORM = 1     # i.i.d. uniform 10-bit org and pred
KIND_ZERO_RESI = 2   # pred == org  (all-zero residual channel)
KIND_SATURATED = 3   # org = 1023, pred = 0 on a checkerboard, swapped elsewhere
KIND_FLAT = 4        # constant org, constant pred
KIND_OUT_OF_RANGE = 5  # texture with 2 % of the samples replaced by negative / > 10-bit Pel values (uint16 cast + clip path)
# content the flat-content guard does not see, or sees only in part (round 3; real video: sky, letterbox edges, screen content)
KIND_PARTIAL_FLAT = 6      # texture with an exactly-constant rectangle over 10-12 % of the aligned quads (just UNDER the guard's 1/8)
KIND_ORG_FLAT_PRED_TEX = 7 # constant org, textured pred (one coherent plane only)
KIND_ORG_TEX_PRED_FLAT = 8 # textured org, constant pred
KIND_RAMP = 9              # smooth horizontal / diagonal luma ramps, pred = the ramp shifted by one or two pixels
KIND_DITHER = 10           # constant +- 1 LSB dither in both planes
KIND_LOW_CONTRAST = 11     # texture of amplitude <= 4 on a constant base, pred = org + noise[-2,2]
KIND_FLAT_ZERO_RESI = 12   # constant org and pred == org (zero residual on flat content)
KIND_NATURAL = 13          # natural-image statistics (round 4): 1/f amplitude spectrum, per-patch contrast, sensor noise; pred = the
                           # same scene displaced by a small motion vector, low-passed (interpolation + quantised reference) + noise
KIND_PARTIAL_NEAR_FLAT = 14  # texture with a NEAR-flat band (+-1 LSB dither / amplitude-4 texture on constants, never exactly constant) over
                             # 40-48 % of the quads: just under the guard's threshold for near-flat content (1/2)
KIND_NAMES = {0: "texture", 1: "uniform", 2: "zero_resi", 3: "saturated", 4: "flat", 5: "out_of_range", 6: "partial_flat",
              7: "org_flat_pred_tex", 8: "org_tex_pred_flat", 9: "ramp", 10: "dither", 11: "low_contrast", 12: "flat_zero_resi",
              13: "natural", 14: "partial_near_flat"}


def natural_patches(size: int, n: int, input_seed: int, first: int = 0):
    """n (org, pred) pairs with natural-image statistics -- the content class the flat-content guard's FLAG RATE is quoted on.

    org : a (size + 8)^2 random-phase field with a 1/f amplitude spectrum (power 1/f^2, the classic natural-image law), scaled to a
          per-patch standard deviation drawn log-uniformly from 6 ... 160 ten-bit steps (sky / out-of-focus background ... foliage),
          around a mean of 120 ... 900, + sensor noise of +-1 step, clipped to 10 bits; the centre size x size window is the CU.
    pred: the same field displaced by an integer motion vector in [-2, 2]^2 (what a merge / skip candidate points at), smoothed by
          the [1 2 1]^2 / 16 kernel with probability 1/2 (sub-pel interpolation + a quantised reference), + noise of amplitude
          0 ... 6 steps (the reference picture's coding error).
    Deterministic in (input_seed, first + i); float FFTs, so reproducible on one numpy build (probes / bench, not fixtures)."""
    m = size + 8
    org = np.empty((n, size, size), np.int16)
    pred = np.empty((n, size, size), np.int16)
    fy = np.fft.fftfreq(m)[:, None]
    fx = np.fft.rfftfreq(m)[None, :]
    f = np.sqrt(fy * fy + fx * fx)
    f[0, 0] = 1.0
    amp = 1.0 / f
    amp[0, 0] = 0.0
    step = 128
    for i0 in range(0, n, step):
        c = min(step, n - i0)
        base = (first + i0) * m * (m // 2 + 1)
        cnt = c * m * (m // 2 + 1)
        ph = uniform(input_seed, "nat/phase", cnt, base).reshape(c, m, m // 2 + 1) * (2.0 * math.pi)
        mag = np.sqrt(-2.0 * np.log(1.0 - uniform(input_seed, "nat/mag", cnt, base).reshape(c, m, m // 2 + 1)))
        field = np.fft.irfft2(amp[None] * mag * np.exp(1j * ph), s=(m, m))
        field -= field.mean(axis=(1, 2), keepdims=True)
        field /= field.std(axis=(1, 2), keepdims=True) + 1e-12
        u = uniform(input_seed, "nat/par", c * 4, (first + i0) * 4).reshape(c, 4)
        sd = 6.0 * np.exp(u[:, 0] * math.log(160.0 / 6.0))
        mean = 120.0 + 780.0 * u[:, 1]
        scene = mean[:, None, None] + sd[:, None, None] * field
        sens = randint(input_seed, "nat/sens", c * m * m, -1, 1, (first + i0) * m * m).reshape(c, m, m)
        o_full = np.clip(np.rint(scene) + sens, 0, 1023)
        mv = randint(input_seed, "nat/mv", c * 2, -2, 2, (first + i0) * 2).reshape(c, 2)
        namp = (u[:, 2] * 7.0).astype(np.int64)  # 0 ... 6
        blur = u[:, 3] < 0.5
        pn = raw_u64(input_seed, "nat/pn", c * size * size, (first + i0) * size * size).reshape(c, size, size)
        for k in range(c):
            ref = o_full[k]
            if blur[k]:
                r = ref.copy()
                r[1:-1, 1:-1] = (ref[:-2, :-2] + 2 * ref[:-2, 1:-1] + ref[:-2, 2:] + 2 * ref[1:-1, :-2] + 4 * ref[1:-1, 1:-1] + 2 * ref[1:-1, 2:]
                                 + ref[2:, :-2] + 2 * ref[2:, 1:-1] + ref[2:, 2:] + 8) // 16
                ref = r
            y0, x0 = 4 + int(mv[k, 0]), 4 + int(mv[k, 1])
            a = int(namp[k])
            noise = (pn[k] % np.uint64(2 * a + 1)).astype(np.int64) - a
            org[i0 + k] = o_full[k, 4:4 + size, 4:4 + size].astype(np.int16)
            pred[i0 + k] = np.clip(ref[y0:y0 + size, x0:x0 + size] + noise, 0, 1023).astype(np.int16)
    return org, pred


def make_patches(size: int, n: int, input_seed: int, kind: int = KIND_TEXTURE, first: int = 0):
    """Returns (org, pred) int16 arrays of shape [n, size, size] (dense, stride == size).

    Patch i is a function of (input_seed, first + i) only, so a rank can generate its own shard.
    """
    org = np.empty((n, size, size), dtype=np.int16)
    pred = np.empty((n, size, size), dtype=np.int16)
    px = size * size
    for i in range(n):
        idx = first + i
        tag = f"patch{idx}"
        if kind == KIND_TEXTURE:
            nb = max(size // 16, 1)
            base = randint(input_seed, tag + "/base", nb * nb, 64, 959).reshape(nb, nb)
            base = np.kron(base, np.ones((size // nb, size // nb), dtype=np.int64))
            tex = randint(input_seed, tag + "/tex", px, -48, 48).reshape(size, size)
            o = np.clip(base + tex, 0, 1023)
            noise = randint(input_seed, tag + "/noise", px, -40, 40).reshape(size, size)
            p = np.clip(o + noise, 0, 1023)
        elif kind == KIND_UNIFORM:
            o = randint(input_seed, tag + "/o", px, 0, 1023).reshape(size, size)
            p = randint(input_seed, tag + "/p", px, 0, 1023).reshape(size, size)
        elif kind == KIND_ZERO_RESI:
            o = randint(input_seed, tag + "/o", px, 0, 1023).reshape(size, size)
            p = o.copy()
        elif kind == KIND_SATURATED:
            yy, xx = np.mgrid[0:size, 0:size]
            chk = ((yy // 4 + xx // 4) & 1).astype(np.int64)
            o = chk * 1023
            p = (1 - chk) * 1023
        elif kind == KIND_OUT_OF_RANGE:
            o = randint(input_seed, tag + "/o", px, 0, 1023).reshape(size, size)
            p = np.clip(o + randint(input_seed, tag + "/n", px, -40, 40).reshape(size, size), 0, 1023)
            weird = np.array([-1, -5, -32768, 32767, 1024, 2000, 1023, 0], np.int64)
            for arr, nm in ((o, "/wo"), (p, "/wp")):
                pos = randint(input_seed, tag + nm + "pos", max(px // 50, 1), 0, px - 1)
                arr.reshape(-1)[pos] = weird[randint(input_seed, tag + nm + "val", len(pos), 0, len(weird) - 1)]
        elif kind == KIND_FLAT:
            o = np.full((size, size), int(randint(input_seed, tag + "/o", 1, 0, 1023)[0]), dtype=np.int64)
            p = np.full((size, size), int(randint(input_seed, tag + "/p", 1, 0, 1023)[0]), dtype=np.int64)
        elif kind in (KIND_PARTIAL_FLAT, KIND_ORG_FLAT_PRED_TEX, KIND_ORG_TEX_PRED_FLAT):
            nb = max(size // 16, 1)
            base = randint(input_seed, tag + "/base", nb * nb, 64, 959).reshape(nb, nb)
            base = np.kron(base, np.ones((size // nb, size // nb), dtype=np.int64))
            o = np.clip(base + randint(input_seed, tag + "/tex", px, -48, 48).reshape(size, size), 0, 1023)
            p = np.clip(o + randint(input_seed, tag + "/noise", px, -40, 40).reshape(size, size), 0, 1023)
            co = int(randint(input_seed, tag + "/co", 1, 0, 1023)[0])
            cp = int(randint(input_seed, tag + "/cp", 1, 0, 1023)[0])
            if kind == KIND_PARTIAL_FLAT:
                # rows [r0, r0 + hh) x all columns constant in BOTH planes: hh / size in [10 %, 12.5 %) of the quads
                hh = max((size * (10 + idx % 3)) // 100, 1)
                if hh * 8 >= size:
                    hh = max(size // 8 - 1, 1)
                r0 = int(randint(input_seed, tag + "/r0", 1, 0, size - hh)[0])
                o[r0:r0 + hh, :] = co
                p[r0:r0 + hh, :] = cp
            elif kind == KIND_ORG_FLAT_PRED_TEX:
                o[:, :] = co
            else:
                p[:, :] = cp
        elif kind == KIND_RAMP:
            yy, xx = np.mgrid[0:size, 0:size]
            a = int(randint(input_seed, tag + "/a", 1, 0, 400)[0])
            sx = int(randint(input_seed, tag + "/sx", 1, 1, 4)[0])   # luma steps per pixel (x4 fixed point below)
            sy = int(randint(input_seed, tag + "/sy", 1, 0, 4)[0]) if idx % 2 else 0   # odd patches: diagonal
            o = np.clip(a + (sx * xx + sy * yy) * max(512 // size, 1) // 4, 0, 1023)
            sh = 1 + idx % 2
            p = np.clip(a + (sx * (xx + sh) + sy * (yy + sh)) * max(512 // size, 1) // 4, 0, 1023)
        elif kind == KIND_DITHER:
            co = int(randint(input_seed, tag + "/co", 1, 1, 1022)[0])
            cp = int(randint(input_seed, tag + "/cp", 1, 1, 1022)[0])
            o = co + randint(input_seed, tag + "/do", px, -1, 1).reshape(size, size)
            p = cp + randint(input_seed, tag + "/dp", px, -1, 1).reshape(size, size)
        elif kind == KIND_LOW_CONTRAST:
            co = int(randint(input_seed, tag + "/co", 1, 8, 1015)[0])
            o = co + randint(input_seed, tag + "/tex", px, -4, 4).reshape(size, size)
            p = np.clip(o + randint(input_seed, tag + "/noise", px, -2, 2).reshape(size, size), 0, 1023)
        elif kind == KIND_FLAT_ZERO_RESI:
            o = np.full((size, size), int(randint(input_seed, tag + "/o", 1, 0, 1023)[0]), dtype=np.int64)
            p = o.copy()
        elif kind == KIND_PARTIAL_NEAR_FLAT:
            nb = max(size // 16, 1)
            base = randint(input_seed, tag + "/base", nb * nb, 64, 959).reshape(nb, nb)
            base = np.kron(base, np.ones((size // nb, size // nb), dtype=np.int64))
            o = np.clip(base + randint(input_seed, tag + "/tex", px, -48, 48).reshape(size, size), 0, 1023)
            p = np.clip(o + randint(input_seed, tag + "/noise", px, -40, 40).reshape(size, size), 0, 1023)
            hh = max((size * (40 + 4 * (idx % 3))) // 100, 1)   # 40 / 44 / 48 % of the rows
            r0 = int(randint(input_seed, tag + "/r0", 1, 0, size - hh)[0])
            co = int(randint(input_seed, tag + "/co", 1, 8, 1015)[0])
            cp = int(randint(input_seed, tag + "/cp", 1, 8, 1015)[0])
            amp = 1 if idx % 2 == 0 else 4   # dither / low contrast
            o[r0:r0 + hh, :] = co + randint(input_seed, tag + "/do", hh * size, -amp, amp).reshape(hh, size)
            p[r0:r0 + hh, :] = cp + randint(input_seed, tag + "/dp", hh * size, -amp, amp).reshape(hh, size)
            # (not a single exactly-constant quad: keep the band clear of the guard's 1/8 rule for exactly flat content)
        elif kind == KIND_NATURAL:
            oo, pp = natural_patches(size, 1, input_seed, idx)
            o, p = oo[0].astype(np.int64), pp[0].astype(np.int64)
        else:
            raise ValueError(kind)
        org[i] = o.astype(np.int16)
        pred[i] = p.astype(np.int16)
    return org, pred


def make_scalars(n: int, input_seed: int, first: int = 0):
    """poc uniform [0,600], qp uniform [17,47] as int32 arrays of length n."""
    poc = randint(input_seed, "poc", n, 0, 600, offset=first).astype(np.int32)
    qp = randint(input_seed, "qp", n, 17, 47, offset=first).astype(np.int32)
    return poc, qp


def make_patches_bulk(size: int, n: int, input_seed: int, first: int = 0):
    """Vectorised KIND_TEXTURE-like generator for large batches (bench workload): same distribution as
    make_patches(kind=KIND_TEXTURE) but drawn from bulk streams, so values differ from make_patches."""
    px = size * size
    nb = max(size // 16, 1)

    def fast(stream, count, lo, hi, offset):  # multiply-shift range reduction on 32 random bits
        r = (raw_u64(input_seed, stream, count, offset) >> np.uint64(32)) * np.uint64(hi - lo + 1)
        return (r >> np.uint64(32)).astype(np.int16) + np.int16(lo)

    org = np.empty((n, size, size), np.int16)
    pred = np.empty((n, size, size), np.int16)
    step = 256
    for i0 in range(0, n, step):  # bounded temporaries
        c = min(step, n - i0)
        f = first + i0
        base = fast("bulk/base", c * nb * nb, 64, 959, f * nb * nb).reshape(c, nb, 1, nb, 1)
        base = np.broadcast_to(base, (c, nb, size // nb, nb, size // nb)).reshape(c, size, size)
        o = np.clip(base + fast("bulk/tex", c * px, -48, 48, f * px).reshape(c, size, size), 0, 1023)
        org[i0:i0 + c] = o
        pred[i0:i0 + c] = np.clip(o + fast("bulk/noise", c * px, -40, 40, f * px).reshape(c, size, size), 0, 1023)
    return org, pred


def make_mix_bulk(size: int, n: int, input_seed: int, flat_frac: float = 0.0, natural: bool = False, first: int = 0):
    """Bench batches beyond the plain texture workload: a fraction `flat_frac` of the CUs (every k-th, evenly spread) replaced by the
    content the flat-content guard re-evaluates exactly -- constant, +-1 LSB dither, ramps, low contrast, in turn -- and / or the
    natural-statistics class instead of the texture class for the rest.  Returns (org, pred, is_flat[n])."""
    org, pred = (natural_patches(size, n, input_seed, first) if natural else make_patches_bulk(size, n, input_seed, first))
    is_flat = np.zeros(n, bool)
    if flat_frac > 0:
        kinds = (KIND_FLAT, KIND_DITHER, KIND_RAMP, KIND_LOW_CONTRAST)
        pos = np.unique(np.floor(np.arange(int(round(n * flat_frac))) / max(flat_frac, 1e-9)).astype(np.int64))
        pos = pos[pos < n]
        for j, i in enumerate(pos):
            o, p = make_patches(size, 1, input_seed ^ 0x5A5A, kinds[j % 4], first + int(i))
            org[i], pred[i] = o[0], p[0]
            is_flat[i] = True
    return org, pred, is_flat


def flat_quad_fraction(org: np.ndarray, pred: np.ndarray, flat_range: int = 6, return_exact: bool = False):
    """Host restatement of the flat-content guard's statistic (csrc/mlt_kernels.hip: quad_near_flat): per CU, the fraction of aligned
    4-pixel quads that are coherent in BOTH planes the network sees (org clipped to 10 bits, |org - pred| clipped) -- range <= flat_range
    or linear to within one step.  The guard flags a CU when the fraction reaches 1/8."""
    o = np.clip(org.astype(np.uint16).astype(np.int64), 0, 1023)
    r = np.clip(np.abs(org.astype(np.uint16).astype(np.int64) - pred.astype(np.uint16).astype(np.int64)), 0, 1023)
    n, S, _ = o.shape
    res = np.ones((n, S, S // 4), bool)
    exact = np.ones((n, S, S // 4), bool)
    for a in (o, r):
        q = a.reshape(n, S, S // 4, 4)
        rng = q.max(-1) - q.min(-1)
        d1 = np.abs(q[..., 0] + q[..., 2] - 2 * q[..., 1])
        d2 = np.abs(q[..., 1] + q[..., 3] - 2 * q[..., 2])
        res &= (rng <= flat_range) | (np.maximum(d1, d2) <= 1)
        exact &= (rng == 0) | (np.maximum(d1, d2) == 0)
    if return_exact:
        return res.reshape(n, -1).mean(1), exact.reshape(n, -1).mean(1)
    return res.reshape(n, -1).mean(1)
