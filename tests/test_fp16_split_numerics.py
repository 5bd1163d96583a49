"""The contraction's operand format, checked without a GPU: NumPy emulation of the two-piece fp16 split (csrc/odf.hip, gemm3_body H2)
against a float64 contraction, next to the exact three-piece bf16 split and a plain f32 chain.  The kernels themselves are
compared on the GPU (tests/test_gpu_odf.py, tools/gemm_accuracy.py); this file pins the arithmetic the design rests on."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def _pieces_bf16(x):
    a1 = _bf16(x); r = (x - a1).astype(np.float32); a2 = _bf16(r); a3 = _bf16((r - a2).astype(np.float32))
    return a1, a2, a3


def _pieces_f16(x):
    h = x.astype(np.float16).astype(np.float32)
    l = (x - h).astype(np.float32).astype(np.float16).astype(np.float32)
    return h, l


def _contract(prods, shape, accum=np.float32):
    """piece products are exact in f32; the MFMA adds a 16-frame block of them to an f32 accumulator"""
    acc = np.zeros(shape, accum)
    K = prods[0][0].shape[1]
    for k0 in range(0, K, 16):
        for ap, sp in prods:
            acc = (acc + ap[:, k0:k0 + 16].astype(np.float64) @ sp[k0:k0 + 16].astype(np.float64)).astype(accum)
    return acc.astype(np.float64)


def _case():
    import fibers_jl_amd as fj
    from fibers_jl_amd import phantom
    from oracle import oracle_np as onp
    bval, bvec = phantom.scheme_gqi()
    sph = fj.sphere_642
    W = onp.gqi_work(bval, bvec, np.asarray(sph.vertices), np.asarray(sph.faces), 1.25)
    A = next(v for v in (W.values() if isinstance(W, dict) else vars(W).values()) if isinstance(v, np.ndarray) and v.ndim == 2 and 270 in v.shape)
    A = np.ascontiguousarray(A, np.float32)
    if A.shape[0] == 270:
        A = A.T.copy()
    rng = np.random.default_rng(7)
    S0 = rng.uniform(800, 1200, size=(1, 512))
    S = (S0 * np.exp(-rng.uniform(0, 6, size=(270, 512))) + rng.normal(0, 20, size=(270, 512))).astype(np.float32)
    return A, np.maximum(S, 0).astype(np.float32)


def _errors(A, S):
    ref = A.astype(np.float64) @ S.astype(np.float64)
    top = np.abs(ref).max(0)
    out = {}
    out["f32"] = (A @ S).astype(np.float64)
    a, s = _pieces_bf16(A), _pieces_bf16(S)
    out["bf16x3"] = _contract([(a[2], s[0]), (a[1], s[1]), (a[0], s[2]), (a[1], s[0]), (a[0], s[1]), (a[0], s[0])], ref.shape)
    # H2: matrix scaled by a power of two into [2^8, 2^9), every voxel's samples by 2^k with the maximum in [2^6, 2^7)
    sa = np.float32(2.0 ** (8 - np.floor(np.log2(np.abs(A).max()))))
    sk = (2.0 ** (6 - np.floor(np.log2(S.max(0))))).astype(np.float32)
    ah, sh = _pieces_f16(A * sa), _pieces_f16(S * sk[None, :])
    p3 = [(ah[1], sh[0]), (ah[0], sh[1]), (ah[0], sh[0])]
    out["fp16x2"] = _contract(p3, ref.shape) / sa / sk[None, :]
    out["fp16x2_scheme_only"] = _contract(p3, ref.shape, np.float64) / sa / sk[None, :]      # f64 accumulate: the format's own error
    return {k: np.abs(v - ref) / top for k, v in out.items()}


def test_two_fp16_pieces_are_as_accurate_as_the_exact_split():
    A, S = _case()
    e = _errors(A, S)
    rms = {k: float(np.sqrt((v * v).mean())) for k, v in e.items()}
    mx = {k: float(v.max()) for k, v in e.items()}
    # the format's own error (23-bit operands, the a_l s_l term dropped) is far below what f32 accumulation adds to ANY of the three
    assert mx["fp16x2_scheme_only"] < 1.5e-7 and rms["fp16x2_scheme_only"] < 3e-8, (mx, rms)
    assert rms["fp16x2_scheme_only"] < 0.2 * rms["bf16x3"]
    # and with the f32 accumulator it is no worse than the exact split and better than the plain f32 chain
    assert rms["fp16x2"] <= 1.05 * rms["bf16x3"] and rms["fp16x2"] < rms["f32"], rms
    assert mx["fp16x2"] < 3e-6 and mx["bf16x3"] < 3e-6 and mx["f32"] < 5e-6, mx


def test_the_sample_scale_keeps_small_samples_and_is_exact_under_powers_of_two():
    A, S = _case()
    S = S.copy()
    S[5:40, :] *= np.float32(2.0 ** -15)                      # samples 2^-15 of the voxel maximum: the low piece is near fp16's subnormals
    e = _errors(A, S)
    assert e["fp16x2_scheme_only"].max() < 1.5e-7
    # power-of-two homogeneity of the split while the low piece is a normal number (the kernel test scales whole volumes: there the
    # voxel's own 2^k undoes the factor before the split)
    x = np.random.default_rng(3).uniform(1.0, 128.0, size=4096).astype(np.float32)
    h1, l1 = _pieces_f16(x)
    h2, l2 = _pieces_f16(x * np.float32(4.0))
    assert np.array_equal(h1 * 4, h2) and np.array_equal(l1 * 4, l2)
    assert np.abs(x - h1 - l1).max() <= 2.0 ** -23 * 128.0                     # 23 of the 24 significant bits
