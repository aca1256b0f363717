"""numpy MODEL of the incremental exact decode the HIP path implements  --  TEST INFRASTRUCTURE, NOT PRODUCT (like oracle/dctts_ref.py:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

The reference re-runs the whole Text2Mel graph at every step (synthesize.py:47-54).
The HIP decode instead keeps
  * TextEnc K,V (computed once: pure function of L),
  * per-layer AudioEnc activation histories indexed by absolute time (Q[t] for t<=j depends only
    on Y[<t], which is final -> exactly incremental),
  * and at every step re-evaluates the 3-key windowed attention and the AudioDec dependency cone
    (85/83/45/15/5/3/1 rows) with the CURRENT window, because the reference tiles step j's mask
    over all time rows (networks.py:145).
All buffers carry PAD zero rows in front of t=0: post-LN activations are never written there, so
reading them reproduces the per-layer causal zero padding (modules.py:121-125,173-177).

This file states that algorithm in numpy so that the cone tables in dc_tts_amd/layers.py and the
buffer scheme can be checked against the oracle's full-recompute loop on CPU, before any GPU run.
"""
import numpy as np

from dc_tts_amd.layers import audioenc_layers, audiodec_layers, audiodec_cone
from oracle import dctts_ref as O

PAD = 64


def _layer_rows(l, P, inbuf, t_rows, dtype):
    """Evaluate one causal layer for absolute-time rows ``t_rows`` (all >= 0) by tap gather from
    ``inbuf`` (B, PAD+T, Cin).  Returns (B, len(t_rows), Cout)."""
    t = np.asarray(t_rows)
    if l.kind == "C":
        Wk = P[l.scope + "/conv1d/kernel"]
        y = inbuf[:, PAD + t, :] @ Wk[0] + P[l.scope + "/conv1d/bias"]
        y = O.normalize(y, P[l.scope + "/normalize/gamma"], P[l.scope + "/normalize/beta"])
        return O.relu(y) if l.act == "relu" else y
    Wk = P[l.scope + "/conv1d/kernel"]
    k = Wk.shape[0]
    y = 0
    for j in range(k):                      # causal: tap j sees x[t - (k-1-j)*rate]
        y = y + inbuf[:, PAD + t - (k - 1 - j) * l.rate, :] @ Wk[j]
    y = y + P[l.scope + "/conv1d/bias"]
    C = l.cout
    H1 = O.sigmoid(O.normalize(y[..., :C], P[l.scope + "/H1/gamma"], P[l.scope + "/H1/beta"]))
    H2 = O.normalize(y[..., C:], P[l.scope + "/H2/gamma"], P[l.scope + "/H2/beta"])
    return H1 * H2 + (1.0 - H1) * inbuf[:, PAD + t, :]


def incremental_decode(L, W, hp, dtype=np.float32, frozen_R=False):
    """Returns (Y (B,T,80), max_att trajectory (B,T) int64).  ``frozen_R=True`` is the 'obvious'
    cache (R[t] frozen at step t, single AudioDec row) that does NOT match the reference."""
    B, T = L.shape[0], hp.max_T
    K, V = O.TextEnc(L, W, hp, dtype)
    ae, ad = audioenc_layers(hp), audiodec_layers(hp)
    cone = audiodec_cone(hp)
    Pe = O._Scoped(W, "Text2Mel/AudioEnc", dtype)
    Pd = O._Scoped(W, "Text2Mel/AudioDec", dtype)
    Ypad = np.zeros((B, PAD + T + 1, hp.n_mels), dtype)          # Ypad[PAD + t] = S[t] = Y[t-1]
    AE = [np.zeros((B, PAD + T, l.cout), dtype) for l in ae]
    Rbuf = np.zeros((B, PAD + T, 2 * hp.d), dtype)
    AD = [np.zeros((B, PAD + T, l.cout), dtype) for l in ad]
    Y = np.zeros((B, T, hp.n_mels), dtype)
    traj = np.zeros((B, T), np.int64)
    p = np.zeros((B,), np.int64)
    scale = dtype(1.0 / np.sqrt(dtype(hp.d)))
    for j in range(T):
        # --- AudioEnc: one new row per utterance
        src = Ypad
        for li, l in enumerate(ae):
            AE[li][:, PAD + j, :] = _layer_rows(l, Pe, src, [j], dtype)[:, 0, :]
            src = AE[li]
        Qh = AE[-1]
        # --- windowed attention for the rows AudioDec C_1 must emit
        offs = cone[0] if not frozen_R else [0]
        rows = [j + o for o in offs if j + o >= 0]
        for b in range(B):
            n0 = int(p[b]); n1 = min(n0 + hp.attention_win_size, hp.max_N)
            q = Qh[b, PAD + np.asarray(rows), :]                 # (R, d)
            lg = (q @ K[b, n0:n1, :].T) * scale                  # (R, <=3)
            lg = lg - lg.max(-1, keepdims=True)
            a = np.exp(lg); a = a / a.sum(-1, keepdims=True)
            Rbuf[b, PAD + np.asarray(rows), :hp.d] = a @ V[b, n0:n1, :]
            Rbuf[b, PAD + np.asarray(rows), hp.d:] = q
            if True:
                jj = rows.index(j)
                traj[b, j] = n0 + int(np.argmax(a[jj]))
        # --- AudioDec dependency cone
        src = Rbuf
        for li, l in enumerate(ad):
            offs_l = cone[li] if not frozen_R else [0]
            rows_l = [j + o for o in offs_l if j + o >= 0]
            AD[li][:, PAD + np.asarray(rows_l), :] = _layer_rows(l, Pd, src, rows_l, dtype)
            src = AD[li]
        Y[:, j, :] = O.sigmoid(AD[-1][:, PAD + j, :])
        Ypad[:, PAD + j + 1, :] = Y[:, j, :]
        p = traj[:, j].copy()
    return Y, traj


def incremental_decode_v3(L, W, hp, dtype=np.float32, stats=None, hc2_rowop=True):
    """numpy MODEL of the round-2 decode data flow (decode3_kernels.h / dctts_api.hip: decode v3).  Same arithmetic as
    ``incremental_decode`` up to fp32 re-association, organised the way the HIP path computes it:

    * AudioDec C_1 never sees R.  With W1 = [W_top ; W_bot] (networks.py:167-174 applied to R = [A.V ; Q], :150-151)
        C_1pre[t] = bias + sum_k a_k(t) * VW[p+k] + C1Q[t],   VW[n] = V[n] . W_top  (once per batch),
                                                              C1Q[t] = Q[t] . W_bot (once per frame, window-independent)
      so re-evaluating C_1 over the 84 older cone rows with a new window is a row operation, not a GEMM.
    * every causal k=3 layer's newest row is  presum + x[t] . W[2]  where  presum = bias + x[t-2d] . W[0] + x[t-d] . W[1]
      only reads rows that are final (AudioEnc) or computed by the bulk of the same frame (AudioDec cone rows < j):
      the two older taps leave the latency-critical chain.
    ``stats`` (dict) receives the minimum top-2 gap of the newest row's window logits."""
    B, T, d = L.shape[0], hp.max_T, hp.d
    K, V = O.TextEnc(L, W, hp, dtype)
    ae, ad = audioenc_layers(hp), audiodec_layers(hp)
    cone = audiodec_cone(hp)
    Pe = O._Scoped(W, "Text2Mel/AudioEnc", dtype)
    Pd = O._Scoped(W, "Text2Mel/AudioDec", dtype)
    W1 = Pd["C_1/conv1d/kernel"][0]                              # (2d, d)
    VW = V @ W1[:d]                                               # (B, N, d)
    b1 = Pd["C_1/conv1d/bias"]
    g1, be1 = Pd["C_1/normalize/gamma"], Pd["C_1/normalize/beta"]
    W2 = Pd["HC_2/conv1d/kernel"]                                 # (3, d, 2d)
    Wt = g1[None, :, None] * W2                                   # diag(gamma1) W2[q]
    VWW = np.einsum('bnc,qcm->bnqm', VW, Wt)                     # (B, N, 3, 2d), once per batch
    b1Wt = np.einsum('c,qcm->qm', b1, Wt); cs = Wt.sum(1); betaW = np.einsum('c,qcm->qm', be1, W2)
    b2v = Pd["HC_2/conv1d/bias"]
    C1QW = np.zeros((B, PAD + T, 3, 2 * d), dtype)
    SC = np.zeros((B, PAD + T, 2 + hp.attention_win_size), dtype)   # per C_1 cone row: mean, rstd of the pre-norm row, attention weights
    Ypad = np.zeros((B, PAD + T + 1, hp.n_mels), dtype)
    AE = [np.zeros((B, PAD + T, l.cout), dtype) for l in ae]
    C1Q = np.zeros((B, PAD + T, d), dtype)
    AD = [np.zeros((B, PAD + T, l.cout), dtype) for l in ad]
    Y = np.zeros((B, T, hp.n_mels), dtype)
    traj = np.zeros((B, T), np.int64)
    p = np.zeros((B,), np.int64)
    scale = dtype(1.0 / np.sqrt(dtype(d)))
    min_gap = np.inf

    def hc_from_pre(l, P, pre, xres):
        C = l.cout
        H1 = O.sigmoid(O.normalize(pre[..., :C], P[l.scope + "/H1/gamma"], P[l.scope + "/H1/beta"]))
        H2 = O.normalize(pre[..., C:], P[l.scope + "/H2/gamma"], P[l.scope + "/H2/beta"])
        return H1 * H2 + (1.0 - H1) * xres

    def presum(l, P, inbuf, t):
        Wk = P[l.scope + "/conv1d/kernel"]
        return P[l.scope + "/conv1d/bias"] + inbuf[:, PAD + t - 2 * l.rate, :] @ Wk[0] + inbuf[:, PAD + t - l.rate, :] @ Wk[1]

    def attn_weights(b, rows):
        n0 = int(p[b]); n1 = min(n0 + hp.attention_win_size, hp.max_N)
        q = AE[-1][b, PAD + np.asarray(rows), :]
        lg = (q @ K[b, n0:n1, :].T) * scale
        e = np.exp(lg - lg.max(-1, keepdims=True))
        return n0, n1, lg, e / e.sum(-1, keepdims=True)

    for j in range(T):
        # ---- chain: AudioEnc row j (presums read final history rows only)
        src = Ypad
        for li, l in enumerate(ae):
            if l.kind == "C":
                AE[li][:, PAD + j, :] = _layer_rows(l, Pe, src, [j], dtype)[:, 0, :]
            else:
                pre = presum(l, Pe, src, j) + src[:, PAD + j, :] @ Pe[l.scope + "/conv1d/kernel"][2]
                AE[li][:, PAD + j, :] = hc_from_pre(l, Pe, pre, src[:, PAD + j, :])
            src = AE[li]
        # ---- chain: attention row j -> next window, C_1 presum; C_1's GEMM is Q[j] . W_bot only
        C1Q[:, PAD + j, :] = AE[-1][:, PAD + j, :] @ W1[d:]
        pre0 = np.zeros((B, d), dtype)
        for b in range(B):
            n0, n1, lg, a = attn_weights(b, [j])
            traj[b, j] = n0 + int(np.argmax(a[0]))                # argmax over the post-softmax row (networks.py:149)
            if n1 - n0 > 1:
                s = np.sort(lg[0].astype(np.float64)); min_gap = min(min_gap, s[-1] - s[-2])
            pre0[b] = b1 + a[0] @ VW[b, n0:n1, :]
        # ---- bulk of frame j: cone rows at offsets < 0 with the CURRENT window (p = prev_max fed at step j)
        rows = [j + o for o in cone[0] if o < 0 and j + o >= 0]
        if rows:
            for b in range(B):
                n0, n1, lg, a = attn_weights(b, rows)
                pre = b1 + a @ VW[b, n0:n1, :] + C1Q[b, PAD + np.asarray(rows), :]
                AD[0][b, PAD + np.asarray(rows), :] = O.normalize(pre, Pd["C_1/normalize/gamma"], Pd["C_1/normalize/beta"])
                mrow = pre.mean(-1); vrow = ((pre - mrow[:, None]) ** 2).mean(-1)
                SC[b, PAD + np.asarray(rows), 0] = mrow; SC[b, PAD + np.asarray(rows), 1] = 1.0 / np.sqrt(vrow + dtype(O.LN_EPS))
                SC[b, PAD + np.asarray(rows), 2:2 + a.shape[1]] = a
        # HC_2 over its cone rows is a row operation too: its input x1[t'] = (pre[t'] - m) r gamma1 + beta1 is affine in pre[t'],
        # and pre[t'] . (diag(gamma1) W2[q]) splits into cached pieces: VWW[n][q] (per batch), C1QW[t'][q] (once per frame)
        l2 = ad[1]
        rows2 = [j + o for o in cone[1] if o < 0 and j + o >= 0]
        if rows2 and hc2_rowop:
            C1QW[:, PAD + j - 1] = np.einsum('bc,qcn->bqn', C1Q[:, PAD + j - 1, :], Wt)       # the newest cached row (computed at the start of the bulk piece)
            for b in range(B):
                n0 = int(p[b]); n1 = min(n0 + hp.attention_win_size, hp.max_N)
                for t in rows2:
                    acc = b2v.copy()
                    for q in range(3):
                        tp = t - (2 - q) * l2.rate
                        if tp < 0:
                            continue
                        m_, r_, a_ = SC[b, PAD + tp, 0], SC[b, PAD + tp, 1], SC[b, PAD + tp, 2:2 + (n1 - n0)]
                        acc = acc + betaW[q] + r_ * (b1Wt[q] + a_ @ VWW[b, n0:n1, q, :] + C1QW[b, PAD + tp, q, :] - m_ * cs[q])
                    AD[1][b, PAD + t, :] = hc_from_pre(l2, Pd, acc, AD[0][b, PAD + t, :])
        for li in range(1 + (1 if hc2_rowop else 0), len(ad)):
            rows_l = [j + o for o in cone[li] if o < 0 and j + o >= 0]
            if rows_l:
                AD[li][:, PAD + np.asarray(rows_l), :] = _layer_rows(ad[li], Pd, AD[li - 1], rows_l, dtype)
        # ---- chain: AudioDec row j
        AD[0][:, PAD + j, :] = O.normalize(pre0 + C1Q[:, PAD + j, :], Pd["C_1/normalize/gamma"], Pd["C_1/normalize/beta"])
        for li in range(1, len(ad)):
            l = ad[li]
            if l.kind == "C":
                AD[li][:, PAD + j, :] = _layer_rows(l, Pd, AD[li - 1], [j], dtype)[:, 0, :]
            else:
                pre = presum(l, Pd, AD[li - 1], j) + AD[li - 1][:, PAD + j, :] @ Pd[l.scope + "/conv1d/kernel"][2]
                AD[li][:, PAD + j, :] = hc_from_pre(l, Pd, pre, AD[li - 1][:, PAD + j, :])
        Y[:, j, :] = O.sigmoid(AD[-1][:, PAD + j, :])
        Ypad[:, PAD + j + 1, :] = Y[:, j, :]
        p = traj[:, j].copy()
    if stats is not None:
        stats["min_top2_logit_gap"] = float(min_gap)
    return Y, traj
