"""CPU check of the packed MFMA-fragment layout (nice_layout.h) without a GPU:
a numpy emulation of v_mfma_f32_16x16x4_f32's lane mapping runs the decoder
chain exactly as csrc/nice_render.hip does (same fragment indices) from the
index table built by the library's HOST function ``xrd_nice_pack_index`` and is
compared with the oracle MLP.  Also checks the transposed fragments used by the
backward data path."""
import ctypes as C

import numpy as np
import pytest
import torch

import nice_oracle as no
from xrdslam_amd import _lib
from xrdslam_amd.engine import nice as en

L = np.arange(64)
Q, M = L >> 4, L & 15


def mfma(a, b, d):
    """d[r][lane] (4x64) += A(16x4)·B(4x16): A[i][k]=a[k*16+i], B[k][j]=b[k*16+j],
    D[row=(l>>4)*4+r][col=l&15]"""
    A = a.reshape(4, 16).T  # [i][k]
    B = b.reshape(4, 16)  # [k][j]
    Dm = A @ B  # [16][16]
    out = d.copy()
    for r in range(4):
        out[r] += Dm[Q * 4 + r, M]
    return out


def kmap(s, q):
    return 16 * (s >> 2) + 4 * q + (s & 3)


def pack(kind, flat):
    lib = _lib.lib()
    k = en.DEC_KINDS[kind]
    n = lib.xrd_nice_pack_len(k)
    idx = np.empty(n, dtype=np.int32)
    assert lib.xrd_nice_pack_index(k, idx.ctypes.data_as(C.c_void_p)) == 0
    ext = np.concatenate([flat, [0.0]]).astype(np.float32)
    return ext[np.where(idx < 0, len(flat), idx)], idx


def offsets(CD):
    KC, KTC = CD // 4, CD // 16
    o = {}
    o['W0'] = 0
    o['W3E'] = o['W0'] + 2 * 24 * 64
    o['WH'] = o['W3E'] + 2 * 24 * 64
    o['WC'] = o['WH'] + 4 * 1024
    o['B'] = o['WC'] + 5 * 2 * KC * 64
    o['BC'] = o['B'] + 160
    o['EMB'] = o['BC'] + 160
    o['WOUT'] = o['EMB'] + 384
    o['BOUT'] = o['WOUT'] + 128
    o['WHT'] = o['BOUT'] + 4
    o['WCT'] = o['WHT'] + 4096
    o['W0T'] = o['WCT'] + 5 * KTC * 512
    o['W3ET'] = o['W0T'] + 3072
    o['LEN'] = o['W3ET'] + 3072
    return o, KC, KTC


def to_dlayout(x):
    """x [16 points, F feats] -> regs [F/16][4][64]: lane(q,i) holds feat 16jt+4q+r"""
    F = x.shape[1]
    out = np.zeros((F // 16, 4, 64), np.float32)
    for jt in range(F // 16):
        for r in range(4):
            out[jt, r] = x[M, 16 * jt + 4 * Q + r]
    return out


def from_dlayout(regs):
    F = regs.shape[0] * 16
    x = np.zeros((16, F), np.float32)
    for jt in range(regs.shape[0]):
        for r in range(4):
            x[M, 16 * jt + 4 * Q + r] = regs[jt, r]
    return x


@pytest.mark.parametrize('kind,CD,OD', [('middle', 32, 1), ('fine', 64, 1),
                                        ('color', 32, 4)])
def test_mlp_chain_emulation(kind, CD, OD):
    torch.manual_seed(3)
    shapes = en.param_shapes(kind)
    sd = {n: torch.randn(*s) * (25.0 if n == 'embedder._B' else 0.3)
          for n, s in shapes}
    flat = en.flatten_state_dict(sd, kind).numpy()
    assert len(flat) == _lib.lib().xrd_nice_flat_len(en.DEC_KINDS[kind])
    pk, _ = pack(kind, flat)
    o, KC, KTC = offsets(CD)
    assert o['LEN'] == len(pk)
    p = (torch.rand(16, 3) * 2 - 1)
    c = torch.randn(16, CD)
    ref = no.mlp_forward(sd, p, c).numpy()

    pn = p.numpy().astype(np.float32)
    cD = to_dlayout(c.numpy())
    acc = np.zeros((2, 4, 64), np.float32)
    acc3 = np.zeros((2, 4, 64), np.float32)
    for jt in range(2):
        for r in range(4):
            acc[jt, r] = pk[o['B'] + 0 * 32 + 16 * jt + 4 * Q + r]
            acc3[jt, r] = pk[o['B'] + 3 * 32 + 16 * jt + 4 * Q + r]
    for s in range(24):
        k = 4 * s + Q
        bk = pk[o['EMB'] + k[:, None] * 4 + np.arange(3)[None]]  # [64,3]
        e = np.sin((pn[M] * bk).sum(1)).astype(np.float32)
        for jt in range(2):
            a0 = pk[o['W0'] + (jt * 24 + s) * 64 + L]
            a3 = pk[o['W3E'] + (jt * 24 + s) * 64 + L]
            acc[jt] = mfma(a0, e, acc[jt])
            acc3[jt] = mfma(a3, e, acc3[jt])
    h = None
    for i in range(5):
        cc = np.zeros((2, 4, 64), np.float32)
        for jt in range(2):
            for r in range(4):
                cc[jt, r] = pk[o['BC'] + i * 32 + 16 * jt + 4 * Q + r]
        for s in range(KC):
            for jt in range(2):
                a = pk[o['WC'] + i * 2 * KC * 64 + (jt * KC + s) * 64 + L]
                cc[jt] = mfma(a, cD[s >> 2, s & 3], cc[jt])
        h = np.maximum(acc, 0) + cc
        if i < 4:
            for jt in range(2):
                for r in range(4):
                    acc[jt, r] = acc3[jt, r] if i + 1 == 3 else \
                        pk[o['B'] + (i + 1) * 32 + 16 * jt + 4 * Q + r]
            for s in range(8):
                for jt in range(2):
                    a = pk[o['WH'] + i * 1024 + (jt * 8 + s) * 64 + L]
                    acc[jt] = mfma(a, h[s >> 2, s & 3], acc[jt])
    hx = from_dlayout(h)  # [16,32]
    wout = pk[o['WOUT']:o['WOUT'] + 128].reshape(4, 32)[:OD]
    out = hx @ wout.T + pk[o['BOUT']:o['BOUT'] + OD]
    np.testing.assert_allclose(out, ref, rtol=2e-3, atol=2e-3)

    # transposed fragments: g_in = W^T g_out in D layout
    g = np.random.default_rng(0).standard_normal((16, 32)).astype(np.float32)
    gD = to_dlayout(g)
    for i in range(1, 5):
        W = sd[f'pts_linears.{i}.weight'].numpy()
        Wh = W[:, 93:] if i == 3 else W
        gp = np.zeros((2, 4, 64), np.float32)
        for kt in range(2):
            for s in range(8):
                a = pk[o['WHT'] + (i - 1) * 1024 + (kt * 8 + s) * 64 + L]
                gp[kt] = mfma(a, gD[s >> 2, s & 3], gp[kt])
        np.testing.assert_allclose(from_dlayout(gp), g @ Wh, rtol=1e-4,
                                   atol=1e-4)
    for i in range(5):
        Wc = sd[f'fc_c.{i}.weight'].numpy()
        gc = np.zeros((KTC, 4, 64), np.float32)
        for kt in range(KTC):
            for s in range(8):
                a = pk[o['WCT'] + i * KTC * 512 + (kt * 8 + s) * 64 + L]
                gc[kt] = mfma(a, gD[s >> 2, s & 3], gc[kt])
        np.testing.assert_allclose(from_dlayout(gc), g @ Wc, rtol=1e-4,
                                   atol=1e-4)
    # embedding transposed: lane group q, reg r of tile kt <-> feature 4(4kt+r)+q
    for key, W in (('W0T', sd['pts_linears.0.weight'].numpy()),
                   ('W3ET', sd['pts_linears.3.weight'].numpy()[:, :93])):
        ge = np.zeros((6, 4, 64), np.float32)
        for kt in range(6):
            for s in range(8):
                a = pk[o[key] + (kt * 8 + s) * 64 + L]
                ge[kt] = mfma(a, gD[s >> 2, s & 3], ge[kt])
        full = g @ W  # [16,93]
        for kt in range(6):
            for r in range(4):
                k = 4 * (4 * kt + r) + Q
                want = np.where(k < 93, full[M, np.minimum(k, 92)], 0.0)
                np.testing.assert_allclose(ge[kt, r], want, rtol=1e-4,
                                           atol=1e-4)


def test_noxyz_chain_emulation():
    torch.manual_seed(4)
    sd = {n: torch.randn(*s) * 0.3 for n, s in en.param_shapes('coarse')}
    flat = en.flatten_state_dict(sd, 'coarse').numpy()
    pk, _ = pack('coarse', flat)
    c = torch.randn(16, 32)
    ref = no.mlp_no_xyz_forward(sd, c).numpy()
    cD = to_dlayout(c.numpy())
    h = cD.copy()
    B, WOUT, BOUT = 6 * 1024, 6 * 1024 + 160, 6 * 1024 + 160 + 32
    for i in range(5):
        ks = 16 if i == 3 else 8
        w = i * 1024 if i <= 3 else 5 * 1024
        acc = np.zeros((2, 4, 64), np.float32)
        for jt in range(2):
            for r in range(4):
                acc[jt, r] = pk[B + i * 32 + 16 * jt + 4 * Q + r]
        for s in range(ks):
            for jt in range(2):
                a = pk[w + (jt * ks + s) * 64 + L]
                src = cD if (i == 3 and s < 8) else h
                acc[jt] = mfma(a, src[(s & 7) >> 2, s & 3], acc[jt])
        h = np.maximum(acc, 0)
    out = from_dlayout(h) @ pk[WOUT:WOUT + 32] + pk[BOUT]
    np.testing.assert_allclose(out, ref[:, 0], rtol=2e-3, atol=2e-3)
    # transposed
    WT = BOUT + 4
    g = np.random.default_rng(1).standard_normal((16, 32)).astype(np.float32)
    gD = to_dlayout(g)
    for i in range(5):
        W = sd[f'pts_linears.{i}.weight'].numpy()
        kts = 4 if i == 3 else 2
        wt = WT + (i * 1024 if i <= 3 else 5 * 1024)
        gp = np.zeros((kts, 4, 64), np.float32)
        for kt in range(kts):
            for s in range(8):
                a = pk[wt + (kt * 8 + s) * 64 + L]
                gp[kt] = mfma(a, gD[s >> 2, s & 3], gp[kt])
        np.testing.assert_allclose(from_dlayout(gp), g @ W, rtol=1e-4,
                                   atol=1e-4)


@pytest.mark.parametrize('kind', ['coarse', 'middle', 'fine', 'color'])
def test_repack_of_a_zero_tailed_decoder_is_the_same_gather(kind):
    """engine/nice.seat_on_zero_tail: a decoder's flat parameter re-seated on
    a buffer with a trailing zero packs to the same fragments with ONE gather
    (no fill, no concatenation), follows in-place updates of the parameter
    (what the fused Adam does), writes into the caller's buffer and is
    idempotent; a parameter that has left the buffer falls back to the
    concatenation."""
    n = _lib.lib().xrd_nice_flat_len(en.DEC_KINDS[kind])
    torch.manual_seed(3)
    flat = torch.nn.Parameter(torch.randn(n))
    ref = en.pack_decoder(flat, kind).clone()
    before = flat.detach().clone()
    old_ptr = flat.data_ptr()
    en.seat_on_zero_tail(flat)
    assert flat.data_ptr() != old_ptr and torch.equal(flat.detach(), before)
    ext = en._zero_tailed(flat)
    assert ext is not None and float(ext[-1]) == 0.0
    got = en.pack_decoder(flat, kind)
    assert torch.equal(got, ref)
    with torch.no_grad():
        flat.mul_(-0.5)
    out = en.pack_decoder(flat, kind, out=got)
    assert out.data_ptr() == got.data_ptr() and torch.equal(out, ref * -0.5)
    assert float(ext[-1]) == 0.0
    en.seat_on_zero_tail(flat)                       # idempotent
    assert en._zero_tailed(flat) is ext
    flat.data = flat.detach().clone()                # moved away
    assert en._zero_tailed(flat) is None
    assert torch.equal(en.pack_decoder(flat, kind), ref * -0.5)
