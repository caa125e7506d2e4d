"""Thin torch-tensor wrappers over the C ABI (one Python function per entry point of klab_mm.h).

Used by the parity tests and by host-side orchestration; tensors only provide device memory
(`data_ptr()`) and the current HIP stream -- the arithmetic is in libklab_mm.so.
"""
import ctypes as C

import torch

from . import _lib as L


def _seed_ptr(seed):
    return None if seed is None else seed.data_ptr()


def gemm(A, B, C_out, *, M, N, K, a_kmajor=True, b_kmajor=True, lda=None, ldb=None, ldc=None, alpha=1.0,
         alpha_dev=None, bias=None, act=L.ACT_NONE, aux=None, aux_mode=L.AUX_NONE, aux_scale=1.0, residual=None,
         drop_p=0.0, seed=None, tag=0, accumulate=False, atomic_ok=False, name_tag=0):
    lib = L.load()
    a = L.GemmArgs()
    a.atomic_ok, a.name_tag = int(atomic_ok), name_tag
    a.M, a.N, a.K = M, N, K
    a.dtype = L.dtype_code(A.dtype)
    assert B.dtype == A.dtype
    a.A, a.lda, a.a_kmajor = A.data_ptr(), (lda if lda is not None else A.stride(0)), int(a_kmajor)
    a.B, a.ldb, a.b_kmajor = B.data_ptr(), (ldb if ldb is not None else B.stride(0)), int(b_kmajor)
    a.C, a.ldc, a.c_dtype = C_out.data_ptr(), (ldc if ldc is not None else C_out.stride(0)), L.dtype_code(C_out.dtype)
    a.accumulate = int(accumulate)
    a.alpha = alpha
    a.alpha_dev = L.ptr(alpha_dev)
    a.bias = L.ptr(bias)
    a.act = act
    a.aux, a.ldaux, a.aux_mode, a.aux_scale = L.ptr(aux), (aux.stride(0) if aux is not None else 0), aux_mode, aux_scale
    a.residual = L.ptr(residual)
    a.ldr = residual.stride(0) if residual is not None else 0
    a.r_dtype = L.dtype_code(residual.dtype) if residual is not None else 0
    a.drop_p, a.seed_dev, a.drop_tag = drop_p, _seed_ptr(seed), tag
    L.check(lib.klab_gemm(C.byref(a), L.stream_ptr()), "klab_gemm")
    return C_out


def quant_fp8_rows(x):
    """bf16 [M, K] -> (uint8 e4m3 [M, K], f32 row scales [M])"""
    import torch
    lib = L.load()
    M, K = x.shape
    x8 = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    sc = torch.empty(M, dtype=torch.float32, device=x.device)
    L.check(lib.klab_quant_fp8_rows(x.data_ptr(), x.stride(0), M, K, x8.data_ptr(), x8.stride(0), sc.data_ptr(), L.stream_ptr()),
            "klab_quant_fp8_rows")
    return x8, sc


def gemm_fp8(A8, sa, B8, sb, C_out, *, alpha=1.0, bias=None, act=L.ACT_NONE, residual=None, drop_p=0.0, seed=None, tag=0, name_tag=0):
    """C[M,N] = epilogue(alpha * sa[m] sb[n] * A8 @ B8^T): A8 [M,K], B8 [N,K] e4m3 bytes (uint8), per-row scales"""
    lib = L.load()
    a = L.GemmArgs()
    a.M, a.K = A8.shape
    a.N = B8.shape[0]
    a.dtype = L.BF16
    a.name_tag = name_tag  # 2: the block-scaled kernel (mmf8.hip) when K % 128 == 0, 3: never
    a.A, a.lda, a.a_kmajor = A8.data_ptr(), A8.stride(0), 1
    a.B, a.ldb, a.b_kmajor = B8.data_ptr(), B8.stride(0), 1
    a.C, a.ldc, a.c_dtype = C_out.data_ptr(), C_out.stride(0), L.dtype_code(C_out.dtype)
    a.alpha = alpha
    a.bias = L.ptr(bias)
    a.act = act
    a.residual = L.ptr(residual)
    a.ldr = residual.stride(0) if residual is not None else 0
    a.r_dtype = L.dtype_code(residual.dtype) if residual is not None else 0
    a.drop_p, a.seed_dev, a.drop_tag = drop_p, _seed_ptr(seed), tag
    L.check(lib.klab_gemm_fp8(C.byref(a), sa.data_ptr(), sb.data_ptr(), 1, L.stream_ptr()), "klab_gemm_fp8")
    return C_out


def rmsnorm_fwd(x, w, y=None, y_f32=None, rstd=None, eps=1e-6, grp=0, grp_stride=0, off=0, drop_p=0.0, seed=None, tag=0):
    lib = L.load()
    rows, d = x.shape
    L.check(lib.klab_rmsnorm_fwd(x.data_ptr(), w.data_ptr(), L.ptr(y), L.dtype_code(y.dtype) if y is not None else 0,
                                 L.ptr(y_f32), L.ptr(rstd), rows, d, eps, grp, grp_stride, off, drop_p, _seed_ptr(seed), tag,
                                 L.stream_ptr()), "klab_rmsnorm_fwd")


def rmsnorm_bwd(dy, x, w, rstd, dres=None, dx=None, dxt=None, dw=None, grp=0, grp_stride=0, off=0, p_y=0.0, tag_y=0,
                p_prev=0.0, tag_prev=0, seed=None):
    lib = L.load()
    rows, d = x.shape
    L.check(lib.klab_rmsnorm_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), rstd.data_ptr(), L.ptr(dres), L.ptr(dx), L.ptr(dxt),
                                 L.dtype_code(dxt.dtype) if dxt is not None else 0, L.ptr(dw), rows, d, grp, grp_stride, off,
                                 p_y, tag_y, p_prev, tag_prev, _seed_ptr(seed), L.stream_ptr()), "klab_rmsnorm_bwd")


def layernorm_fwd(y, gamma, beta, shortcut=None, out=None, outt=None, mean=None, rstd=None, eps=1e-5, grp=0, grp_stride=0,
                  off=0, drop_p=0.0, seed=None, tag=0):
    lib = L.load()
    rows, Cc = y.shape
    L.check(lib.klab_layernorm_fwd(y.data_ptr(), L.dtype_code(y.dtype), gamma.data_ptr(), beta.data_ptr(), L.ptr(shortcut),
                                   L.ptr(out), L.ptr(outt), L.dtype_code(outt.dtype) if outt is not None else 0, L.ptr(mean),
                                   L.ptr(rstd), rows, Cc, eps, grp, grp_stride, off, drop_p, _seed_ptr(seed), tag,
                                   L.stream_ptr()), "klab_layernorm_fwd")


def swin_qkv_attn_fused(x, wqkv, bqkv, ctx, bias, logit_scale, *, B, R, w, shift, H, C):
    """ctx = window attention of (x @ wqkv.T + bqkv) without materialising q|k|v (frozen Swin-V2, HF/swinv2:389-455)."""
    lib = L.load()
    L.check(lib.klab_swin_qkv_attn_fused(x.data_ptr(), wqkv.data_ptr(), L.ptr(bqkv), ctx.data_ptr(), bias.data_ptr(), logit_scale.data_ptr(),
                                         L.dtype_code(x.dtype), B, R, w, shift, H, C, L.stream_ptr()), "klab_swin_qkv_attn_fused")


def swin_proj_ln_fused(x, shortcut, w, b, gamma, beta, out, outt=None, eps=1e-5):
    """out = shortcut + LN(x @ w.T + b)*gamma+beta (frozen Swin-V2 attention-output half, C in {64,128})."""
    lib = L.load()
    M, Cc = x.shape
    L.check(lib.klab_swin_proj_ln_fused(x.data_ptr(), shortcut.data_ptr(), w.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                        out.data_ptr(), L.ptr(outt), L.dtype_code(x.dtype), M, Cc, eps, L.stream_ptr()), "klab_swin_proj_ln_fused")


def swin_mlp_fused(x, shortcut, w1, b1, w2, b2, gamma, beta, out, outt=None, eps=1e-5):
    """out = shortcut + LN(fc2(GELU(fc1(x)+b1))+b2)*gamma+beta (frozen Swin-V2 MLP half, C in {64,128}); raises
    NotImplementedError for other widths (HF/swinv2:539-563, 697-702)."""
    lib = L.load()
    M, Cc = x.shape
    L.check(lib.klab_swin_mlp_fused(x.data_ptr(), shortcut.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                    gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), L.ptr(outt), L.dtype_code(x.dtype), M, Cc, eps,
                                    L.stream_ptr()), "klab_swin_mlp_fused")


def rmsnorm_fwd_q8(x, w, y, rstd, y8, yscale, eps=1e-6):
    """fp8 mode: y (bf16) = RMS-norm(x) * w and the same rows in e4m3 (y8 uint8 [rows, d]) with one scale per row"""
    lib = L.load()
    rows, d = x.shape
    L.check(lib.klab_rmsnorm_fwd_q8(x.data_ptr(), w.data_ptr(), y.data_ptr(), L.ptr(rstd), y8.data_ptr(), yscale.data_ptr(), rows, d, eps, 0.0,
                                    None, 0, L.stream_ptr()), "klab_rmsnorm_fwd_q8")


def layernorm_fwd_q8(y, gamma, beta, shortcut, out, outt, mean, rstd, o8, oscale, eps=1e-5):
    """fp8 mode: klab_layernorm_fwd with a bf16 `outt`, plus the same rows in e4m3 and one scale per row"""
    lib = L.load()
    rows, Cc = y.shape
    L.check(lib.klab_layernorm_fwd_q8(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), L.ptr(shortcut), L.ptr(out), outt.data_ptr(), L.ptr(mean),
                                      L.ptr(rstd), o8.data_ptr(), oscale.data_ptr(), rows, Cc, eps, L.stream_ptr()), "klab_layernorm_fwd_q8")


def gelu_fwd(z, a):
    """a = gelu(z), elementwise (the trainable Swin tower keeps the pre-activation z for the backward)"""
    lib = L.load()
    L.check(lib.klab_gelu_fwd(z.data_ptr(), a.data_ptr(), L.dtype_code(z.dtype), z.numel(), L.stream_ptr()), "klab_gelu_fwd")


def gelu_fwd_q8(z, a, a8, ascale):
    lib = L.load()
    rows, F = z.shape
    L.check(lib.klab_gelu_fwd_q8(z.data_ptr(), a.data_ptr(), a8.data_ptr(), ascale.data_ptr(), rows, F, L.stream_ptr()), "klab_gelu_fwd_q8")


def layernorm_bwd(dout, y, gamma, mean, rstd, dy=None, dgamma=None, dbeta=None, grp=0, grp_stride=0, off=0, drop_p=0.0,
                  seed=None, tag=0, dprev_bias=None):
    lib = L.load()
    rows, Cc = y.shape
    if dprev_bias is not None:
        L.check(lib.klab_layernorm_bwd_bias(dout.data_ptr(), y.data_ptr(), L.dtype_code(y.dtype), gamma.data_ptr(), mean.data_ptr(),
                                            rstd.data_ptr(), L.ptr(dy), L.ptr(dgamma), L.ptr(dbeta), dprev_bias.data_ptr(), rows, Cc, grp,
                                            grp_stride, off, drop_p, _seed_ptr(seed), tag, L.stream_ptr()), "klab_layernorm_bwd_bias")
        return
    L.check(lib.klab_layernorm_bwd(dout.data_ptr(), y.data_ptr(), L.dtype_code(y.dtype), gamma.data_ptr(), mean.data_ptr(),
                                   rstd.data_ptr(), L.ptr(dy), L.ptr(dgamma), L.ptr(dbeta), rows, Cc, grp, grp_stride, off,
                                   drop_p, _seed_ptr(seed), tag, L.stream_ptr()), "klab_layernorm_bwd")


def _attn_args(q, k, v, ctx, lse, bias, causal, B, H, Lq, Lk, dk, drop_p, seed, tag, ldq, ldk, ldv, ldo):
    a = L.AttnArgs()
    a.dtype = L.dtype_code(q.dtype)
    a.q, a.ldq = q.data_ptr(), ldq
    a.k, a.ldk = k.data_ptr(), ldk
    a.v, a.ldv = v.data_ptr(), ldv
    a.bias, a.causal = L.ptr(bias), int(causal)
    a.ctx, a.ldo, a.lse = ctx.data_ptr(), ldo, L.ptr(lse)
    a.B, a.H, a.Lq, a.Lk, a.dk = B, H, Lq, Lk, dk
    a.drop_p, a.seed_dev, a.drop_tag = drop_p, _seed_ptr(seed), tag
    return a


def t5_attn_fwd(q, k, v, ctx, lse, *, B, H, Lq, Lk, dk, bias=None, causal=False, drop_p=0.0, seed=None, tag=0,
                ldq=None, ldk=None, ldv=None, ldo=None):
    lib = L.load()
    a = _attn_args(q, k, v, ctx, lse, bias, causal, B, H, Lq, Lk, dk, drop_p, seed, tag,
                   ldq or q.stride(0), ldk or k.stride(0), ldv or v.stride(0), ldo or ctx.stride(0))
    L.check(lib.klab_t5_attn_fwd(C.byref(a), L.stream_ptr()), "klab_t5_attn_fwd")


def swin_linear_ln_fused(x, shortcut, w, bias, gamma, beta, out, outt=None, *, eps=1e-5):
    """shortcut + LayerNorm(x @ w.T + bias) * gamma + beta in one launch (klab_swin_linear_ln_fused); NotImplementedError outside the
    envelope (bf16, 256 output columns, K a multiple of 64)"""
    lib = L.load()
    M, K = x.shape
    L.check(lib.klab_swin_linear_ln_fused(x.data_ptr(), shortcut.data_ptr(), w.data_ptr(), bias.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                          out.data_ptr(), L.ptr(outt), L.dtype_code(x.dtype), M, K, w.shape[0], eps, L.stream_ptr()),
            "klab_swin_linear_ln_fused")


def swin_patch_embed_fused(pixels, w_padded, bias, gamma, beta, out, outt, *, patch=4, eps=1e-5):
    """LayerNorm(Conv2d(3 -> C, k 4, s 4)(pixels)) in one launch (klab_swin_patch_embed_fused); w_padded: bf16 [C, 64], columns 48.. zero"""
    lib = L.load()
    B, cin, H, _ = pixels.shape
    L.check(lib.klab_swin_patch_embed_fused(pixels.data_ptr(), w_padded.data_ptr(), w_padded.stride(0), bias.data_ptr(), gamma.data_ptr(),
                                            beta.data_ptr(), out.data_ptr(), L.ptr(outt), L.dtype_code(w_padded.dtype), B, cin, H, patch,
                                            out.shape[1], eps, L.stream_ptr()), "klab_swin_patch_embed_fused")


def t5_attn_fused_fwd(x, gamma, w, xn, rstd, proj, ctx, lse, *, B, H, Lq, Lk, dk, eps=1e-6, bias=None, causal=False, cross=False, k=None, v=None,
                      ldk=None, ldv=None, drop_p=0.0, seed=None, tag=0):
    """T5LayerNorm -> q|k|v (self) / q (cross) projection -> attention in one launch (klab_t5_attn_fused_fwd); NotImplementedError
    outside the envelope (bf16, d_model 512, head dim 64, at most 64 queries / keys)"""
    lib = L.load()
    fa = L.AttnFusedArgs()
    fa.x, fa.gamma, fa.eps, fa.d_model = x.data_ptr(), gamma.data_ptr(), eps, x.shape[1]
    fa.w, fa.xn, fa.rstd, fa.proj, fa.ldproj, fa.cross = w.data_ptr(), xn.data_ptr(), rstd.data_ptr(), proj.data_ptr(), proj.stride(0), int(cross)
    kk, vv = (k, v) if cross else (proj, proj)
    fa.attn = _attn_args(proj, kk, vv, ctx, lse, bias, causal, B, H, Lq, Lk, dk, drop_p, seed, tag, proj.stride(0),
                         ldk or kk.stride(0), ldv or vv.stride(0), ctx.stride(0))
    L.check(lib.klab_t5_attn_fused_fwd(C.byref(fa), L.stream_ptr()), "klab_t5_attn_fused_fwd")


def t5_attn_bwd(q, k, v, ctx, lse, dctx, dq, dk_out, dv, *, B, H, Lq, Lk, dk, bias=None, causal=False, dbias=None,
                drop_p=0.0, seed=None, tag=0, ldq=None, ldk=None, ldv=None, ldo=None, lddo=None, lddq=None, lddk=None,
                lddv=None, ds_ws=None):
    lib = L.load()
    a = _attn_args(q, k, v, ctx, lse, bias, causal, B, H, Lq, Lk, dk, drop_p, seed, tag,
                   ldq or q.stride(0), ldk or k.stride(0), ldv or v.stride(0), ldo or ctx.stride(0))
    a.dctx, a.lddo = dctx.data_ptr(), lddo or dctx.stride(0)
    a.dq, a.lddq = dq.data_ptr(), lddq or dq.stride(0)
    a.dk_out, a.lddk = dk_out.data_ptr(), lddk or dk_out.stride(0)
    a.dv, a.lddv = dv.data_ptr(), lddv or dv.stride(0)
    a.dbias = L.ptr(dbias)
    a.ds_ws = L.ptr(ds_ws)
    L.check(lib.klab_t5_attn_bwd(C.byref(a), L.stream_ptr()), "klab_t5_attn_bwd")


def _swin_args(qkv, ctx, bias, logit_scale, lse, B, R, w, shift, H, Cc, bias_table=None, v_bias=None, dv_bias=None):
    a = L.SwinAttnArgs()
    a.v_bias, a.dv_bias = L.ptr(v_bias), L.ptr(dv_bias)
    a.dtype = L.dtype_code(qkv.dtype)
    a.qkv, a.ctx, a.bias, a.logit_scale, a.lse = qkv.data_ptr(), ctx.data_ptr(), L.ptr(bias), logit_scale.data_ptr(), L.ptr(lse)
    a.bias_table = L.ptr(bias_table)
    a.B, a.R, a.w, a.shift, a.H, a.C = B, R, w, shift, H, Cc
    return a


def swin_attn_fwd(qkv, ctx, bias, logit_scale, lse=None, *, B, R, w, shift, H, C, bias_table=None, mfma=True, v_bias=None):
    """bias [H, n, n] dense, or bias=None + bias_table [(2w-1)^2, H] (large windows: looked up per score); mfma=True hands the
    kernel the scratch its matrix-core form needs for windows of more than 64 tokens (False: the vector-ALU tiled kernel)"""
    import torch
    lib = L.load()
    a = _swin_args(qkv, ctx, bias, logit_scale, lse, B, R, w, shift, H, C, bias_table, v_bias)
    if mfma and (w * w > 64 or R % w):
        nbytes = lib.klab_swin_attn_bwd_ws_bytes(a.dtype, B, R, w, H, C)
        if nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=qkv.device)
            a.bwd_ws, a.bwd_ws_bytes = ws.data_ptr(), nbytes
    L.check(lib.klab_swin_attn_fwd(C_byref(a), L.stream_ptr()), "klab_swin_attn_fwd")


def swin_attn_bwd(qkv, ctx, bias, logit_scale, lse, dctx, dqkv, dbias=None, dlogit_scale=None, *, B, R, w, shift, H, C, mfma=True,
                  bias_table=None, dbias_table=None, v_bias=None, dv_bias=None):
    """mfma=True hands the kernel its scratch (when the shape is inside the matrix-core envelope); False forces the VALU form."""
    import torch
    lib = L.load()
    a = _swin_args(qkv, ctx, bias, logit_scale, lse, B, R, w, shift, H, C, bias_table, v_bias, dv_bias)
    a.dbias_table = L.ptr(dbias_table)
    a.dctx, a.dqkv, a.dbias, a.dlogit_scale = dctx.data_ptr(), dqkv.data_ptr(), L.ptr(dbias), L.ptr(dlogit_scale)
    nbytes = lib.klab_swin_attn_bwd_ws_bytes(a.dtype, B, R, w, H, C) if mfma else 0
    if nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=qkv.device)
        a.bwd_ws, a.bwd_ws_bytes = ws.data_ptr(), nbytes
    L.check(lib.klab_swin_attn_bwd(C_byref(a), L.stream_ptr()), "klab_swin_attn_bwd")


C_byref = C.byref


def swin_cpb_bias(coords, index, w0, b0, w2, table, bias, hidden=None, *, n, heads):
    lib = L.load()
    L.check(lib.klab_swin_cpb_bias(coords.data_ptr(), index.data_ptr(), w0.data_ptr(), b0.data_ptr(), w2.data_ptr(),
                                   table.data_ptr(), L.ptr(hidden), bias.data_ptr(), coords.shape[0], n, heads, w0.shape[0],
                                   L.stream_ptr()), "klab_swin_cpb_bias")


def swin_cpb_table(coords, w0, b0, w2, table, bias_table, hidden=None, *, heads):
    lib = L.load()
    L.check(lib.klab_swin_cpb_table(coords.data_ptr(), w0.data_ptr(), b0.data_ptr(), w2.data_ptr(), table.data_ptr(), L.ptr(hidden),
                                    bias_table.data_ptr(), coords.shape[0], heads, w0.shape[0], L.stream_ptr()), "klab_swin_cpb_table")


def swin_cpb_table_bwd(dbias_table, bias_table, coords, hidden, w2, dtable, dw0, db0, dw2, *, heads):
    lib = L.load()
    L.check(lib.klab_swin_cpb_table_bwd(dbias_table.data_ptr(), bias_table.data_ptr(), coords.data_ptr(), hidden.data_ptr(), w2.data_ptr(),
                                        dtable.data_ptr(), dw0.data_ptr(), db0.data_ptr(), dw2.data_ptr(), coords.shape[0], heads,
                                        hidden.shape[1], L.stream_ptr()), "klab_swin_cpb_table_bwd")


def ce_fwd(logits, labels, inv_n, loss_row, loss, write_grad=True):
    lib = L.load()
    rows, V = logits.shape
    L.check(lib.klab_ce_fwd(logits.data_ptr(), logits.stride(0), L.dtype_code(logits.dtype), labels.data_ptr(), rows, V,
                            inv_n.data_ptr(), loss_row.data_ptr(), loss.data_ptr(), int(write_grad), L.stream_ptr()), "klab_ce_fwd")


def embed_fwd(ids, table, out, *, shift_right=False, L_seq=1, start_id=0, pad_id=0, drop_p=0.0, seed=None, tag=0, err=None):
    lib = L.load()
    rows, d = out.shape
    L.check(lib.klab_embed_fwd(ids.data_ptr(), int(shift_right), L_seq, start_id, pad_id, table.data_ptr(), table.shape[0],
                               out.data_ptr(), rows, d, drop_p, _seed_ptr(seed), tag, L.ptr(err), L.stream_ptr()), "klab_embed_fwd")


def embed_bwd(ids, dh, dtable, *, shift_right=False, L_seq=1, start_id=0, pad_id=0, drop_p=0.0, seed=None, tag=0):
    lib = L.load()
    rows, d = dh.shape
    L.check(lib.klab_embed_bwd(ids.data_ptr(), int(shift_right), L_seq, start_id, pad_id, dh.data_ptr(), dtable.data_ptr(),
                               dtable.shape[0], rows, d, drop_p, _seed_ptr(seed), tag, L.stream_ptr()), "klab_embed_bwd")


def relbias_fwd(table, bucket, bias):
    lib = L.load()
    H, Lq, Lk = bias.shape
    L.check(lib.klab_relbias_fwd(table.data_ptr(), bucket.data_ptr(), bias.data_ptr(), H, Lq, Lk, L.stream_ptr()), "klab_relbias_fwd")


def relbias_bwd(dbias, bucket, dtable):
    lib = L.load()
    H, Lq, Lk = dbias.shape
    L.check(lib.klab_relbias_bwd(dbias.data_ptr(), bucket.data_ptr(), dtable.data_ptr(), H, Lq, Lk, dtable.shape[0],
                                 L.stream_ptr()), "klab_relbias_bwd")


def im2col_patch(pixels, out, P):
    lib = L.load()
    B, Cin, Himg, _ = pixels.shape
    ldo = out.shape[-1]
    if ldo == Cin * P * P:
        L.check(lib.klab_im2col_patch(pixels.data_ptr(), out.data_ptr(), L.dtype_code(out.dtype), B, Cin, Himg, P, L.stream_ptr()),
                "klab_im2col_patch")
    else:  # padded rows: tail columns are zero-filled
        L.check(lib.klab_im2col_patch_ld(pixels.data_ptr(), out.data_ptr(), L.dtype_code(out.dtype), B, Cin, Himg, P, ldo, L.stream_ptr()),
                "klab_im2col_patch_ld")


def merge_gather(x, out, *, B, R, C):
    lib = L.load()
    L.check(lib.klab_merge_gather(x.data_ptr(), out.data_ptr(), L.dtype_code(out.dtype), B, R, C, L.stream_ptr()), "klab_merge_gather")


def merge_scatter(dm, dx, *, B, R, C):
    lib = L.load()
    L.check(lib.klab_merge_scatter(dm.data_ptr(), dx.data_ptr(), B, R, C, L.stream_ptr()), "klab_merge_scatter")


def colsum(dy, out):
    lib = L.load()
    M, N = dy.shape
    L.check(lib.klab_colsum(dy.data_ptr(), dy.stride(0), L.dtype_code(dy.dtype), M, N, out.data_ptr(), L.stream_ptr()), "klab_colsum")


def convert(x, y, scale=1.0):
    lib = L.load()
    L.check(lib.klab_convert(x.data_ptr(), y.data_ptr(), L.dtype_code(y.dtype), x.numel(), scale, L.stream_ptr()), "klab_convert")


def image_preprocess(src_u8, desc, n, max_h, max_w, pixel_values, *, mid=256, out=224, filter_a=3, filter_b=2,
                     rescale=1.0 / 255 / 255, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """klab_image_preprocess: src_u8 device uint8 bytes, desc device int64 [n, 2] rows of {offset, height | width << 32}
    (the klab_image_desc layout), or filter_a=0 with src_u8 = [n, mid, mid, 3]."""
    import torch
    lib = L.load()
    m3 = (C.c_float * 3)(*mean)
    s3 = (C.c_float * 3)(*std)
    ws, nbytes = None, 0
    if filter_a:
        nbytes = lib.klab_image_preprocess_ws_bytes(n, max_h, mid)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=src_u8.device)
    L.check(lib.klab_image_preprocess(src_u8.data_ptr(), L.ptr(desc), n, max_h, max_w, mid, out, filter_a, filter_b, rescale,
                                      C.cast(m3, C.c_void_p), C.cast(s3, C.c_void_p), pixel_values.data_ptr(), L.ptr(ws), nbytes,
                                      L.stream_ptr()), "klab_image_preprocess")


# ---- JPEG (row f-1): host entropy decoding + device reconstruction ---------------------------------------------------------
def jpeg_read_info(data: bytes):
    """header of one JPEG file (host only): a _lib.JpegInfo"""
    lib = L.load()
    info = L.JpegInfo()
    data = data if isinstance(data, bytes) else bytes(data)
    L.check(lib.klab_jpeg_read_info(C.cast(C.c_char_p(data), C.c_void_p), len(data), C.byref(info)), "klab_jpeg_read_info")
    return info


def jpeg_entropy_decode_batch(datas, n_threads=8):
    """Huffman-decode a list of JPEG byte strings on the host (threaded).  Returns (coefs int16 numpy [total_blocks, 64] in pinned
    memory when CUDA is available, qt uint16 numpy [n, 3, 64], items = (_lib.JpegItem * n) with coef_block0 filled, rgb_off = byte
    offsets of 16-byte aligned HWC RGB images, total RGB bytes).  Unsupported files (CMYK, 12-bit, arithmetic-coded, ...) raise
    NotImplementedError naming the index."""
    import numpy as np
    import torch
    lib = L.load()
    n = len(datas)
    items = (L.JpegItem * n)()
    datas = [d if isinstance(d, bytes) else bytes(d) for d in datas]
    bufs = [C.c_char_p(d) for d in datas]  # pointers into the bytes objects (read-only use, `datas` outlives the calls)
    blocks, rgb = 0, 0
    for i, d in enumerate(datas):
        L.check(lib.klab_jpeg_read_info(C.cast(bufs[i], C.c_void_p), len(d), C.byref(items[i].info)), f"klab_jpeg_read_info[{i}]")
        f = items[i].info
        if not f.supported:
            raise NotImplementedError(f"klab: JPEG {i} is outside the device decoder (progressive={f.progressive}, components={f.ncomp}, "
                                      f"precision={f.precision}, sampling={list(f.hs)}x{list(f.vs)})")
        items[i].coef_block0 = blocks
        items[i].rgb_off = rgb
        blocks += f.coef_blocks
        rgb += (f.width * f.height * 3 + 15) // 16 * 16
    coefs_t = torch.empty((max(blocks, 1), 64), dtype=torch.int16, pin_memory=torch.cuda.is_available())
    coefs = coefs_t.numpy()
    qt = np.zeros((n, 3, 64), np.uint16)
    ptrs = (C.c_void_p * n)(*[C.cast(b, C.c_void_p) for b in bufs])
    sizes = (C.c_size_t * n)(*[len(d) for d in datas])
    cps = (C.c_void_p * n)(*[coefs.ctypes.data + items[i].coef_block0 * 128 for i in range(n)])
    infos = (L.JpegInfo * n)()
    rcs = (C.c_int * n)()
    rc = lib.klab_jpeg_entropy_decode_batch(C.cast(ptrs, C.c_void_p), C.cast(sizes, C.c_void_p), n, C.cast(cps, C.c_void_p),
                                            qt.ctypes.data, C.cast(infos, C.c_void_p), C.cast(rcs, C.c_void_p), int(n_threads))
    if rc:
        bad = [i for i in range(n) if rcs[i]]
        L.check(rcs[bad[0]] if bad else rc, f"klab_jpeg_entropy_decode_batch (images {bad})")
    return coefs_t, qt, items, rgb


def jpeg_decode_device(coefs_t, qt, items, rgb_bytes, device="cuda"):
    """device half: returns (rgb uint8 device buffer holding the HWC images at items[i].rgb_off, desc int64 [n, 2] device tensor in
    the klab_image_desc layout that image_preprocess takes)"""
    import numpy as np
    import torch
    lib = L.load()
    n = len(items)
    dev = torch.device(device)
    coefs_dev = coefs_t.to(dev, non_blocking=True)
    qt_dev = torch.from_numpy(qt.view(np.int16)).to(dev)
    raw = np.frombuffer(bytes(items), dtype=np.uint8).copy()
    items_dev = torch.from_numpy(raw).to(dev)
    nbytes = lib.klab_jpeg_decode_ws_bytes(C.cast(items, C.c_void_p), n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rgb = torch.empty(max(rgb_bytes, 16), dtype=torch.uint8, device=dev)
    L.check(lib.klab_jpeg_decode_device(coefs_dev.data_ptr(), qt_dev.data_ptr(), C.cast(items, C.c_void_p), items_dev.data_ptr(), n,
                                        rgb.data_ptr(), ws.data_ptr(), nbytes, L.stream_ptr()), "klab_jpeg_decode_device")
    desc = np.zeros((n, 2), np.int64)
    for i in range(n):
        desc[i, 0] = items[i].rgb_off
        desc[i, 1] = items[i].info.height | (items[i].info.width << 32)
    # (coefs_dev / qt_dev / items_dev / ws are consumed by kernels already enqueued on the current stream; the caching allocator
    # keeps their memory ordered behind that stream)
    return rgb, torch.from_numpy(desc).to(dev)


def jpeg_decode(datas, device="cuda", n_threads=8, pipelined=False):
    """list of JPEG byte strings -> list of HWC uint8 RGB device tensors (views into one buffer): `Image.open(f).convert('RGB')`"""
    if pipelined:
        rgb, _desc, items = jpeg_decode_pipelined(datas, device, n_threads)
    else:
        coefs_t, qt, items, rgb_bytes = jpeg_entropy_decode_batch(datas, n_threads)
        rgb, _desc = jpeg_decode_device(coefs_t, qt, items, rgb_bytes, device)
    out = []
    for it in items:
        h, w = it.info.height, it.info.width
        out.append(rgb[it.rgb_off:it.rgb_off + h * w * 3].view(h, w, 3))
    return out


def jpeg_decode_pipelined(datas, device="cuda", n_threads=8, n_chunks=4):
    """file bytes -> (rgb device buffer, desc device tensor [n, 2] in the klab_image_desc layout, items).  The batch is cut into
    chunks: while the host threads Huffman-decode chunk i+1, chunk i's coefficients cross PCIe and are reconstructed on the GPU
    (the copies and kernels are asynchronous on the current stream; the host never waits for the device)."""
    import numpy as np
    import torch
    lib = L.load()
    n = len(datas)
    dev = torch.device(device)
    datas = [d if isinstance(d, bytes) else bytes(d) for d in datas]
    items = (L.JpegItem * n)()
    blocks, rgb_bytes = 0, 0
    for i, d in enumerate(datas):
        L.check(lib.klab_jpeg_read_info(C.cast(C.c_char_p(d), C.c_void_p), len(d), C.byref(items[i].info)), f"klab_jpeg_read_info[{i}]")
        f = items[i].info
        if not f.supported:
            raise NotImplementedError(f"klab: JPEG {i} is outside the device decoder (components={f.ncomp}, precision={f.precision}, "
                                      f"sampling={list(f.hs)}x{list(f.vs)})")
        items[i].coef_block0, items[i].rgb_off = blocks, rgb_bytes
        blocks += f.coef_blocks
        rgb_bytes += (f.width * f.height * 3 + 15) // 16 * 16
    coefs_t = torch.empty((max(blocks, 1), 64), dtype=torch.int16, pin_memory=True)
    qt_t = torch.zeros((n, 192), dtype=torch.int16, pin_memory=True)
    coefs, qt = coefs_t.numpy(), qt_t.numpy()
    coefs_dev = torch.empty((max(blocks, 1), 64), dtype=torch.int16, device=dev)
    qt_dev = torch.empty((n, 192), dtype=torch.int16, device=dev)
    items_dev = torch.from_numpy(np.frombuffer(bytes(items), dtype=np.uint8).copy()).to(dev)
    rgb = torch.empty(max(rgb_bytes, 16), dtype=torch.uint8, device=dev)
    isz = C.sizeof(L.JpegItem)
    per = max(1, -(-n // max(1, n_chunks)))
    keep = []
    for lo in range(0, n, per):
        hi = min(n, lo + per)
        m = hi - lo
        ptrs = (C.c_void_p * m)(*[C.cast(C.c_char_p(datas[i]), C.c_void_p) for i in range(lo, hi)])
        sizes = (C.c_size_t * m)(*[len(datas[i]) for i in range(lo, hi)])
        cps = (C.c_void_p * m)(*[coefs.ctypes.data + items[i].coef_block0 * 128 for i in range(lo, hi)])
        rcs = (C.c_int * m)()
        rc = lib.klab_jpeg_entropy_decode_batch(C.cast(ptrs, C.c_void_p), C.cast(sizes, C.c_void_p), m, C.cast(cps, C.c_void_p),
                                                qt.ctypes.data + lo * 384, None, C.cast(rcs, C.c_void_p), int(n_threads))
        if rc:
            bad = [lo + i for i in range(m) if rcs[i]]
            L.check(rcs[bad[0] - lo] if bad else rc, f"klab_jpeg_entropy_decode_batch (images {bad})")
        b0 = items[lo].coef_block0
        b1 = items[hi - 1].coef_block0 + items[hi - 1].info.coef_blocks
        coefs_dev[b0:b1].copy_(coefs_t[b0:b1], non_blocking=True)
        qt_dev[lo:hi].copy_(qt_t[lo:hi], non_blocking=True)
        host_items = C.cast(C.addressof(items) + lo * isz, C.c_void_p)
        nbytes = lib.klab_jpeg_decode_ws_bytes(host_items, m)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        keep.append(ws)
        # qt index of image i in the kernel is (i - lo) * 3 + c: pass the chunk's slice of the table
        L.check(lib.klab_jpeg_decode_device(coefs_dev.data_ptr(), qt_dev.data_ptr() + lo * 384, host_items, items_dev.data_ptr() + lo * isz, m,
                                            rgb.data_ptr(), ws.data_ptr(), nbytes, L.stream_ptr()), "klab_jpeg_decode_device")
    desc = np.zeros((n, 2), np.int64)
    for i in range(n):
        desc[i, 0] = items[i].rgb_off
        desc[i, 1] = items[i].info.height | (items[i].info.width << 32)
    return rgb, torch.from_numpy(desc).to(dev), items
