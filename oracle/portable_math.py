"""TEST INFRASTRUCTURE — CPU oracle, not product code.

Portable fp32 ``expf`` / ``logf`` built only from IEEE-754 correctly rounded
``+ - * /`` and integer bit manipulation, evaluated in a fixed order.  The HIP
sampler kernel (grapes_amd/csrc/sampler_kernels.hip, ``p_expf``/``p_logf``)
performs the *same* operation sequence with fp contraction disabled, so the
Gumbel-top-k keys are bit-identical on the CPU oracle and on gfx950 and the
drawn index sets cannot differ through last-bit rounding.

The polynomial/range-reduction scheme is the classic FreeBSD-msun / musl
single-precision one (public algorithm); it is accurate to <1 ulp, i.e. the keys
agree with torch's ``sigmoid().log() + gumbel`` (reference modules/utils.py:37-42)
to a few ulp, which tests/test_oracle_golden.py pins against the committed
reference fixtures.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.

p_logf / p_expf / _scale2 restate the schemes and constants of FreeBSD msun's
e_logf.c / e_expf.c (as carried by musl).  Those sources carry this notice,
preserved here as it requires:

 * ====================================================
 * Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *
 * Developed at SunPro, a Sun Microsystems, Inc. business.
 * Permission to use, copy, modify, and distribute this
 * software is freely granted, provided that this notice
 * is preserved.
 * ====================================================
"""
import numpy as np

F32 = np.float32
_LN2_HI_L = F32(6.9313812256e-01)   # 0x3f317180
_LN2_LO_L = F32(9.0580006145e-06)   # 0x3717f7d1
_LG1 = F32(0.66666662693)           # 0xaaaaaa.0p-24
_LG2 = F32(0.40000972152)           # 0xccce13.0p-25
_LG3 = F32(0.28498786688)           # 0x91e9ee.0p-25
_LG4 = F32(0.24279078841)           # 0xf89e26.0p-26
_TWO25 = F32(33554432.0)

_LN2_HI_E = F32(6.9314575195e-01)   # 0x3f317200
_LN2_LO_E = F32(1.4286067653e-06)   # 0x35bfbe8e
_INVLN2 = F32(1.4426950216e+00)     # 0x3fb8aa3b
_P1 = F32(1.6666625440e-1)          # 0xaaaa8f.0p-26
_P2 = F32(-2.7667332906e-3)         # -0xb55215.0p-32


def _bits(x):
    return np.ascontiguousarray(x, dtype=F32).view(np.uint32)


def _from_bits(u):
    return np.ascontiguousarray(u, dtype=np.uint32).view(F32)


def p_logf(x):
    """Portable natural log, fp32 in / fp32 out (numpy array)."""
    x = np.ascontiguousarray(x, dtype=F32).copy()
    with np.errstate(all="ignore"):
        ix = _bits(x).copy()
        neg = (ix >> 31) != 0
        zero = (ix & np.uint32(0x7FFFFFFF)) == 0
        nan_or_inf = (ix & np.uint32(0x7FFFFFFF)) >= np.uint32(0x7F800000)
        den = (ix < np.uint32(0x00800000)) & ~zero & ~neg
        k = np.zeros(x.shape, dtype=np.int32)
        k[den] = -25
        x2 = np.where(den, x * _TWO25, x).astype(F32)
        ix = _bits(x2).copy()
        ix = ix + np.uint32(0x3F800000 - 0x3F3504F3)
        k = k + (ix >> 23).astype(np.int32) - 127
        ix = (ix & np.uint32(0x007FFFFF)) + np.uint32(0x3F3504F3)
        xr = _from_bits(ix)
        f = xr - F32(1.0)
        s = f / (F32(2.0) + f)
        z = s * s
        w = z * z
        t1 = w * (_LG2 + w * _LG4)
        t2 = z * (_LG1 + w * _LG3)
        R = t2 + t1
        hfsq = (F32(0.5) * f) * f
        dk = k.astype(F32)
        res = ((((s * (hfsq + R)) + dk * _LN2_LO_L) - hfsq) + f) + dk * _LN2_HI_L
        res = res.astype(F32)
        res = np.where(zero, F32(-np.inf), res)
        res = np.where(neg & ~zero, F32(np.nan), res)
        # +inf -> +inf, NaN -> NaN
        res = np.where(nan_or_inf & ~neg, x + x, res)
        res = np.where(nan_or_inf & neg & np.isnan(x), x, res)
    return res.astype(F32)


def _scale2(y, k):
    """y * 2**k with two exactly specified multiplications (k int32 array)."""
    k = k.astype(np.int32)
    # split k = k1 + k2 with both in the normal exponent range
    k1 = np.clip(k, -100, 100)
    k2 = np.clip(k - k1, -100, 100)
    m1 = _from_bits(((k1 + 127).astype(np.uint32)) << np.uint32(23))
    m2 = _from_bits(((k2 + 127).astype(np.uint32)) << np.uint32(23))
    return ((y * m1).astype(F32) * m2).astype(F32)


def p_expf(x):
    """Portable exp, fp32 in / fp32 out (numpy array)."""
    x = np.ascontiguousarray(x, dtype=F32)
    with np.errstate(all="ignore"):
        hx = _bits(x)
        sign = (hx >> 31).astype(np.int32)
        ax = hx & np.uint32(0x7FFFFFFF)
        isnan = ax > np.uint32(0x7F800000)
        ovf = (ax >= np.uint32(0x42B17218)) & (sign == 0) & ~isnan     # x >= 88.722839
        unf = (ax >= np.uint32(0x42CFF1B5)) & (sign == 1) & ~isnan     # x <= -103.972084
        big = ax > np.uint32(0x3EB17218)        # |x| > 0.5 ln2
        bigger = ax > np.uint32(0x3F851592)     # |x| > 1.5 ln2
        tiny = ax <= np.uint32(0x39000000)      # |x| <= 2**-13 -> 1 + x
        halfs = np.where(sign == 1, F32(-0.5), F32(0.5)).astype(F32)
        kf = (_INVLN2 * x + halfs).astype(F32)
        # C float->int conversion truncates toward zero
        kf = np.where(np.isfinite(kf), kf, F32(0.0))
        kf = np.clip(kf, F32(-200.0), F32(200.0))
        k_big = np.trunc(kf).astype(np.int32)
        k_small = (1 - sign - sign).astype(np.int32)
        k = np.where(bigger, k_big, k_small)
        k = np.where(big, k, 0).astype(np.int32)
        kfl = k.astype(F32)
        hi = np.where(big, x - kfl * _LN2_HI_E, x).astype(F32)
        lo = np.where(big, kfl * _LN2_LO_E, F32(0.0)).astype(F32)
        xr = np.where(big, hi - lo, x).astype(F32)
        xx = xr * xr
        c = xr - xx * (_P1 + xx * _P2)
        y = F32(1.0) + (((xr * c) / (F32(2.0) - c) - lo) + hi)
        y = y.astype(F32)
        res = _scale2(y, k)
        res = np.where(tiny, F32(1.0) + x, res)
        res = np.where(ovf, F32(np.inf), res)
        res = np.where(unf, F32(0.0), res)
        res = np.where(isnan, x, res)
    return res.astype(F32)


def p_sigmoid(l):
    """1 / (1 + exp(-l)) — same form as torch.sigmoid's scalar kernel."""
    l = np.ascontiguousarray(l, dtype=F32)
    with np.errstate(all="ignore"):
        e = p_expf(-l)
        return (F32(1.0) / (F32(1.0) + e)).astype(F32)


# torch.distributions.Gumbel(0,1).sample(): Uniform(finfo.tiny, 1 - finfo.eps)
# then -log(-log(u))  (reference modules/utils.py:40-41).
_TINY = np.finfo(np.float32).tiny
_HI = F32(1.0) - np.finfo(np.float32).eps
_SPAN = F32(_HI - _TINY)


def gumbel_from_uniform(r):
    """r = torch.rand(n) bits -> Gumbel(0,1) noise, op-for-op portable."""
    r = np.ascontiguousarray(r, dtype=F32)
    with np.errstate(all="ignore"):
        u = (r * _SPAN).astype(F32) + F32(_TINY)
        x1 = p_logf(u)
        x3 = p_logf(-x1)
        return (-x3).astype(F32)


def gumbel_keys(logits, r):
    """keys = log(sigmoid(l)) + gumbel(r)   (reference modules/utils.py:42)."""
    with np.errstate(all="ignore"):
        lp = p_logf(p_sigmoid(logits))
        return (lp + gumbel_from_uniform(r)).astype(F32)


def float_order_key(keys):
    """Monotone map fp32 -> uint32 (ascending); +NaN sorts greatest (as torch.topk)."""
    b = _bits(keys)
    neg = (b >> 31) != 0
    return np.where(neg, ~b, b | np.uint32(0x80000000)).astype(np.uint32)


# ---------------------------------------------------------------- Philox4x32-10
_PH_M0 = np.uint64(0xD2511F53)
_PH_M1 = np.uint64(0xCD9E8D57)
_PH_W0 = np.uint32(0x9E3779B9)
_PH_W1 = np.uint32(0xBB67AE85)


def philox_uniform(seed, offset, n):
    """n uniforms in [0,1): element i uses counter (i//4 + offset, 0, 0, 0), key = seed
    (lo, hi), lane i%4, value (x >> 8) * 2**-24.  Same generator as the HIP kernel."""
    n = int(n)
    nblk = (n + 3) // 4
    ctr = np.arange(nblk, dtype=np.uint64) + np.uint64(offset)
    c0 = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    c1 = (ctr >> np.uint64(32)).astype(np.uint32)
    c2 = np.zeros(nblk, dtype=np.uint32)
    c3 = np.zeros(nblk, dtype=np.uint32)
    k0 = np.uint32(seed & 0xFFFFFFFF)
    k1 = np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _PH_M0 * c0.astype(np.uint64)
            p1 = _PH_M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            c0, c1, c2, c3 = (hi1 ^ c1 ^ k0), lo1, (hi0 ^ c3 ^ k1), lo0
            k0 = np.uint32((int(k0) + int(_PH_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_PH_W1)) & 0xFFFFFFFF)
    out = np.stack([c0, c1, c2, c3], axis=1).reshape(-1)[:n]
    return ((out >> np.uint32(8)).astype(F32) * F32(2.0 ** -24)).astype(F32)
