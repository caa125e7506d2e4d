// klab_engine: the whole MyModel.forward / backward (ref/models/model.py:19-26 + autograd) as ONE
// native launch sequence on one HIP stream.
//
// The reference drives this path from Python through three HuggingFace modules, ~2,000 ATen kernel
// launches per step with a host round trip each; here the host side is a C++ plan that, for a fixed
// (B, Ls, Lt), knows every buffer in one caller-owned workspace and issues the hand-written kernels
// back to back (no allocation, no sync => the sequence is hipGraph-capturable).
//   forward : frozen T5 encoder over src ids (ref model.py:20-21)  ->  rows [N_img, Le) of the encoder
//             input;  Swin-V2 over pixels (model.py:22) -> rows [0, N_img)  (the torch.cat of
//             model.py:23 is a row remap in the two final norms);  T5 encoder + decoder + tied LM head
//             + cross-entropy (model.py:26, HF/t5:1009-1054).
//   backward: segments 0 (LM head + decoder + shared embedding), 1 (encoder), 2 (Swin, only when
//             --image_model_train); the caller may launch a gradient all-reduce per segment while the
//             next one runs (DDP overlap, ref/train.py:26,62).
// Parameters stay caller-owned fp32 tensors named exactly like the HuggingFace state dict
// (SURVEY §8b); the engine publishes (name, shape, grad offset) so the host mirrors that schema.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "gemm_shared.h"  // klab::tl_launch_probe: a launch carries the probe's events as its own start / stop events
#include "klab_mm.h"

namespace {

#define RC(x)                 \
  do {                        \
    int rc__ = (x);           \
    if (rc__ != 0) return rc__; \
  } while (0)

struct ParamInfo {
  std::string name;
  std::vector<long> shape;
  long numel = 0;
  long grad_off = -1;   // element offset in the model's flat f32 grad buffer, -1 = frozen
  long warena_off = -1; // element offset in the compute-dtype weight arena (-1: used in fp32 directly)
  long farena_off = -1; // element offset in the f32 side arena (fused bias vectors)
};

struct T5LayerIdx { int q, k, v, o, relb, ln0, cq, ck, cv, co, ln1, wi, wo, ln2; };
struct T5Idx { int shared; std::vector<T5LayerIdx> enc, dec; int enc_final, dec_final; };
struct SwinBlockIdx { int ls, c0w, c0b, c2w, qw, qb, kw, vw, vb, pw, pb, ln1w, ln1b, f1w, f1b, f2w, f2b, ln2w, ln2b; };
struct SwinStageIdx { std::vector<SwinBlockIdx> blk; int redw, mnw, mnb; };
struct SwinIdx { int pew, peb, penw, penb; std::vector<SwinStageIdx> st; int lnw, lnb; };

struct T5LayerBufs {
  void *xn1, *qkv, *ctx; float *lse, *rstd1;
  void *xn2, *qc, *ctx2; float *lse2, *rstd2;
  void *xn3, *hmid; float* rstd3;
};
struct T5StackBufs {
  std::vector<float*> h;
  std::vector<T5LayerBufs> L;
  void* out_t = nullptr; float* rstd_f = nullptr;
  float* bias = nullptr; float* dbias = nullptr; const int* bucket = nullptr;
  int M = 0, Lseq = 0;
};
struct SwinBlockBufs {
  float* x_in; void* xt_in;   // block input (f32 stream + dtype copy)
  void *qkv, *ctx, *po; float *lse, *mean1, *rstd1; float* h1; void* h1t;
  void *z, *a, *fo; float *mean2, *rstd2; float* h2; void* h2t;
  float *bias, *table, *hidden;
  float* btab;  // windows of more than 64 tokens: 16*sigmoid(table) [(2w-1)^2, H], looked up per score (no dense bias)
  int R, w, shift, H, C; long M;
};
struct SwinStageBufs { std::vector<SwinBlockBufs> blk; void *mg, *mo; float *mmean, *mrstd; float* xm; void* xmt; };

struct Bump {
  char* base; size_t off = 0;
  explicit Bump(void* b) : base((char*)b) {}
  void* take(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    void* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
};

}  // namespace

struct klab_engine {
  klab_model_cfg cfg;
  std::vector<ParamInfo> P[3];  // 0 swin, 1 lang, 2 main
  SwinIdx si; T5Idx li, mi;
  long grad_elems[3] = {0, 0, 0};
  long seg_zero_off[3] = {0, 0, 0}, seg_zero_len[3] = {0, 0, 0};  // per backward segment: small atomically-accumulated grads
  long seg_off[3] = {0, 0, 0}, seg_len[3] = {0, 0, 0};            // per segment: extent in its flat grad buffer
  long warena_elems = 0, farena_elems = 0;
  long kvall_w_off = -1, kvall_g_off = -1;  // decoder cross k|v of all layers (weight arena / grad offsets)
  // gradient buckets inside a backward segment, in the order they become final: one per T5 layer (its GEMM weights are
  // adjacent in the flat buffer; final when the layer's grouped weight-gradient launch has run on the side stream), one per
  // Swin block.  The data-parallel reducer all-reduces bucket i behind bucket_ev[seg][i] while the rest of the segment runs.
  struct Bucket { long off, len; };
  std::vector<Bucket> buckets[3];
  std::vector<hipEvent_t> bucket_ev[3];
  bool bucket_ev_live[3] = {false, false, false};
  bool bucket_events_on = false;  // recorded only when a data-parallel reducer asked for them (12 event records per step cost 0.6 %)  // recorded by the last backward of that segment (never under graph replay)
  int pe_k0 = 0, pe_kp = 0;  // patch-embedding weight rows: K0 = in_ch*patch^2 values, stored at a pitch of pe_kp (zero-padded)
  // ---- bound state ----
  bool bound = false;
  int B = 0, Ls = 0, Lt = 0, Le = 0, Nimg = 0;
  size_t es = 4;
  std::vector<const float*> W[3];
  float* G[3] = {nullptr, nullptr, nullptr};
  void* warena = nullptr; float* farena = nullptr;
  // fp8 mode (BASELINE configs[4]): e4m3 copies of the GEMM weights + per-row scales, refreshed from the bf16 arena at the
  // start of every forward; activations are quantised per token in front of each forward Linear GEMM.  Backward stays bf16.
  bool fp8 = false;
  void* w8 = nullptr; float* wscale = nullptr; void* qdesc = nullptr; int n_qdesc = 0; long q_rows = 0;
  void* x8 = nullptr; float* xscale = nullptr; long x8_bytes = 0, xscale_rows = 0;
  void* x8s = nullptr; float* xscales = nullptr; long x8s_bytes = 0, xscales_rows = 0;  // the side stream's own staging (language encoder)
  // fp8 mode: the staging buffer of stream [0 main / 1 side] currently holds the e4m3 rows of THIS bf16 matrix (written by the norm
  // kernel that produced it); consumed by the next forward GEMM on that stream, cleared by anything else that writes the staging
  const void* q8_src[2] = {nullptr, nullptr}; int q8_rows[2] = {0, 0}, q8_k[2] = {0, 0};
  void* cast_desc = nullptr; int n_cast = 0; long cast_total4 = 0;      // trainable GEMM weights (cast every forward)
  void* adam_desc = nullptr; int n_adam = 0; long adam_total4 = 0, adam_split4 = 0;
  // RMS-norm weight gradients of a stack: per-workgroup partials of every norm, folded by one reduction per stack
  float* rms_part = nullptr; long rms_part_stride = 0; float** rms_dst_dev[2] = {nullptr, nullptr}; int rms_ncalls[2] = {0, 0};
  void* cast_desc_frozen = nullptr; int n_cast_frozen = 0; long cast_total4_frozen = 0;  // frozen towers (cast when dirty)
  bool frozen_valid = false;  // arena copies + CPB bias tables of the frozen towers are up to date
  void* fcast_desc = nullptr; int n_fcast = 0; long fcast_total4 = 0;
  uint32_t* seed_dev = nullptr; int* err_dev = nullptr;
  const int *enc_bucket = nullptr, *dec_bucket = nullptr, *lang_bucket = nullptr;
  std::vector<const float*> swin_coords; std::vector<const int*> swin_index; std::vector<int> swin_ntab;
  T5StackBufs lang, enc, dec;
  // greedy decoding with a K/V cache (klab_engine_decode_step): one position per sample, contiguous [B, .] rows
  float* dc_h[2] = {nullptr, nullptr}; float* dc_rstd = nullptr;
  void *dc_xn = nullptr, *dc_q = nullptr, *dc_ctx = nullptr, *dc_hmid = nullptr, *dc_out = nullptr, *dc_logits = nullptr;
  // lang scratch (no grad => reused across layers)
  void* kv_all = nullptr; void* dkv_all = nullptr;
  void* logits = nullptr; float *loss_row = nullptr, *inv_n = nullptr, *loss = nullptr;
  float *dh_a = nullptr, *dh_b = nullptr, *dxn = nullptr, *denc = nullptr;
  void *dctx = nullptr, *ds_ws = nullptr;
  // per-sub-layer gradient operands (no buffer is rewritten inside a backward segment, so the weight-gradient GEMMs
  // can trail on the side stream with read-after-write events only)
  std::vector<void*> dy_pool, dhmid_pool, dqkv_pool, dqc_pool;
  std::vector<hipEvent_t> evpool; int ev_next = 0;
  void *cols = nullptr, *pe_out = nullptr; float *pe_mean = nullptr, *pe_rstd = nullptr; float* x0 = nullptr; void* x0t = nullptr;
  std::vector<SwinStageBufs> sw;
  float *sw_fmean = nullptr, *sw_frstd = nullptr;
  // swin backward scratch
  float *sdh_a = nullptr, *sdh_b = nullptr, *sdm = nullptr; void *sdy = nullptr, *sdctx = nullptr, *sdqkv = nullptr, *sda = nullptr;
  void* sattn_ws = nullptr; size_t sattn_ws_bytes = 0;
  // bias-gradient scratch of the trainable tower: ONE slice per block (dense d(bias) | d(bias table) | d(table)), cleared by a single
  // fill at the head of swin_backward instead of two fills in front of every block's attention backward
  float *sdbias = nullptr, *sdtable = nullptr, *sdbtab = nullptr;
  long sdb_stride = 0, sdt_stride = 0; size_t sdz_bytes = 0;
  // Swin weight gradients on the side stream: per-block gradient operands in two alternating sets (block k+2 reuses set k & 1
  // once swin_done_ev[k & 1] -- recorded on the side stream behind block k's weight gradients -- has fired)
  void *sdyA[2] = {nullptr, nullptr}, *sdyB[2] = {nullptr, nullptr}, *sdaP[2] = {nullptr, nullptr}, *sdqkvP[2] = {nullptr, nullptr};
  hipEvent_t swin_done_ev[2] = {nullptr, nullptr};
  // side stream: independent chains run beside the main one (frozen language encoder || Swin; weight gradients ||
  // the activation-gradient chain); joined back with events before anything the caller can observe
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // backward entry: the gradient slices of segments 0 and 1 and the two attention-bias accumulators are cleared on the SIDE stream
  // (behind one main->side event) while the LM-head input gradient runs; the main stream waits for ev_zero before its first use
  hipEvent_t ev_zero = nullptr;
  bool seg1_zeroed = false;      // segment 1's slice is clear and nothing has written it since
  bool dbias_zeroed[2] = {false, false};  // [decoder stack, encoder stack]
  bool denc_in_dxn = false;      // segment 0 left d(encoder output) in dxn, where segment 1's stack reads it
  float* loss_out = nullptr;     // one-shot destination of the next forward's loss (klab_engine_set_loss_out)
  bool bias_ready[3] = {false, false, false};  // [stack id]: the stack's position bias was computed ahead of it, on the side stream
  bool dec_embed_ready = false;   // ... and the decoder's input embedding
  bool inv_n_ready = false;       // ... and 1 / n_valid of the labels
  bool kv_wgrad_done = false;     // the cross k|v weight gradients went out with the decoder layers' groups
  // stream capture is illegal on the legacy default stream (where PyTorch runs unless told otherwise): in graph mode
  // calls arriving on stream 0 are executed on this engine-owned stream, fenced in and out with events
  hipStream_t own = nullptr; hipEvent_t ev_in = nullptr, ev_out = nullptr;
  // hipGraph replay of the (allocation-free, sync-free) launch sequences: [0..1] forward up to the LM head by
  // training flag, [2..3] cross-entropy by want_grad, [4..6] backward segments.  First use runs eagerly (sets
  // kernel attributes), second use captures, later uses replay.  Rebinding drops them.
  bool use_graph = false;
  struct GraphSlot { hipGraphExec_t exec = nullptr; int uses = 0; bool failed = false; } gs[7];
  // engine-owned copies of the per-step inputs: graph nodes need stable addresses
  float* pixels_buf = nullptr; long long *src_buf = nullptr, *tgt_buf = nullptr; float* dloss_buf = nullptr;
  const float* pixels_cur = nullptr;  // what this forward's patch embedding reads: the caller's tensor, or pixels_buf under graph replay
  uint32_t seed_base = 0; bool seed_set = false;
  // probes: HIP events around selected launches, on the stream they are launched on.  Channel 0: the LM-head logits GEMM of
  // each forward; channel 1: every grouped weight-gradient launch of the T5 backward (side stream)
  bool probe_on = false;
  struct Probe { std::vector<hipEvent_t> a, b; std::vector<double> flops; int n = 0; } probe[2];
  float p_train = 0.f;   // dropout prob in effect for the last forward (0 in eval)
  const long long* last_tgt = nullptr;
};

namespace {

// ------------------------------------------------------------------------------------------------
// parameter tables (names = HuggingFace state-dict keys)
// ------------------------------------------------------------------------------------------------
int add_param(std::vector<ParamInfo>& v, const std::string& name, std::vector<long> shape) {
  ParamInfo p;
  p.name = name; p.shape = shape; p.numel = 1;
  for (long s : shape) p.numel *= s;
  v.push_back(p);
  return (int)v.size() - 1;
}

void build_t5_params(const klab_t5_cfg& c, bool encoder_only, std::vector<ParamInfo>& v, T5Idx& ix) {
  const long d = c.d_model, inner = (long)c.n_heads * c.d_kv, ff = c.d_ff;
  ix.shared = add_param(v, "shared.weight", {c.vocab, d});
  auto stack = [&](const char* pre, int nl, bool dec, std::vector<T5LayerIdx>& L, int& fin) {
    for (int i = 0; i < nl; ++i) {
      T5LayerIdx l;
      memset(&l, -1, sizeof(l));
      std::string bp = std::string(pre) + "block." + std::to_string(i) + ".";
      std::string ap = bp + "layer.0.SelfAttention.";
      l.q = add_param(v, ap + "q.weight", {inner, d});
      l.k = add_param(v, ap + "k.weight", {inner, d});
      l.v = add_param(v, ap + "v.weight", {inner, d});
      l.o = add_param(v, ap + "o.weight", {d, inner});
      if (i == 0) l.relb = add_param(v, ap + "relative_attention_bias.weight", {c.rel_buckets, c.n_heads});
      l.ln0 = add_param(v, bp + "layer.0.layer_norm.weight", {d});
      int li = 1;
      if (dec) {
        std::string cp = bp + "layer.1.EncDecAttention.";
        l.cq = add_param(v, cp + "q.weight", {inner, d});
        l.ck = add_param(v, cp + "k.weight", {inner, d});
        l.cv = add_param(v, cp + "v.weight", {inner, d});
        l.co = add_param(v, cp + "o.weight", {d, inner});
        l.ln1 = add_param(v, bp + "layer.1.layer_norm.weight", {d});
        li = 2;
      }
      std::string fp = bp + "layer." + std::to_string(li) + ".";
      l.wi = add_param(v, fp + "DenseReluDense.wi.weight", {ff, d});
      l.wo = add_param(v, fp + "DenseReluDense.wo.weight", {d, ff});
      l.ln2 = add_param(v, fp + "layer_norm.weight", {d});
      L.push_back(l);
    }
    fin = add_param(v, std::string(pre) + "final_layer_norm.weight", {d});
  };
  if (!encoder_only) stack("decoder.", c.n_dec_layers, true, ix.dec, ix.dec_final);
  stack("encoder.", c.n_layers, false, ix.enc, ix.enc_final);
}

void build_swin_params(const klab_swin_cfg& c, std::vector<ParamInfo>& v, SwinIdx& ix) {
  const long C0 = c.embed_dim;
  ix.pew = add_param(v, "embeddings.patch_embeddings.projection.weight", {C0, c.in_ch, c.patch, c.patch});
  ix.peb = add_param(v, "embeddings.patch_embeddings.projection.bias", {C0});
  ix.penw = add_param(v, "embeddings.norm.weight", {C0});
  ix.penb = add_param(v, "embeddings.norm.bias", {C0});
  for (int s = 0; s < c.n_stages; ++s) {
    SwinStageIdx st;
    const long C = C0 << s, H = c.heads[s], F = (long)c.mlp_ratio * C;
    for (int b = 0; b < c.depths[s]; ++b) {
      SwinBlockIdx k;
      memset(&k, -1, sizeof(k));
      std::string bp = "encoder.layers." + std::to_string(s) + ".blocks." + std::to_string(b) + ".";
      std::string sp = bp + "attention.self.";
      k.ls = add_param(v, sp + "logit_scale", {H, 1, 1});
      k.c0w = add_param(v, sp + "continuous_position_bias_mlp.0.weight", {512, 2});
      k.c0b = add_param(v, sp + "continuous_position_bias_mlp.0.bias", {512});
      k.c2w = add_param(v, sp + "continuous_position_bias_mlp.2.weight", {H, 512});
      k.qw = add_param(v, sp + "query.weight", {C, C});
      if (c.qkv_bias) k.qb = add_param(v, sp + "query.bias", {C});
      k.kw = add_param(v, sp + "key.weight", {C, C});
      k.vw = add_param(v, sp + "value.weight", {C, C});
      if (c.qkv_bias) k.vb = add_param(v, sp + "value.bias", {C});
      k.pw = add_param(v, bp + "attention.output.dense.weight", {C, C});
      k.pb = add_param(v, bp + "attention.output.dense.bias", {C});
      k.ln1w = add_param(v, bp + "layernorm_before.weight", {C});
      k.ln1b = add_param(v, bp + "layernorm_before.bias", {C});
      k.f1w = add_param(v, bp + "intermediate.dense.weight", {F, C});
      k.f1b = add_param(v, bp + "intermediate.dense.bias", {F});
      k.f2w = add_param(v, bp + "output.dense.weight", {C, F});
      k.f2b = add_param(v, bp + "output.dense.bias", {C});
      k.ln2w = add_param(v, bp + "layernorm_after.weight", {C});
      k.ln2b = add_param(v, bp + "layernorm_after.bias", {C});
      st.blk.push_back(k);
    }
    st.redw = st.mnw = st.mnb = -1;
    if (s < c.n_stages - 1) {
      std::string dp = "encoder.layers." + std::to_string(s) + ".downsample.";
      st.redw = add_param(v, dp + "reduction.weight", {2 * C, 4 * C});
      st.mnw = add_param(v, dp + "norm.weight", {2 * C});
      st.mnb = add_param(v, dp + "norm.bias", {2 * C});
    }
    ix.st.push_back(st);
  }
  const long Cl = C0 << (c.n_stages - 1);
  ix.lnw = add_param(v, "layernorm.weight", {Cl});
  ix.lnb = add_param(v, "layernorm.bias", {Cl});
}

inline long pad8(long n) { return (n + 7) & ~7L; }

// weight-arena (compute dtype) offsets; q|k|v and the decoder's cross k|v of ALL layers are adjacent
void plan_arenas(klab_engine* e) {
  long w = 0, f = 0;
  // (16-element alignment: the fp8 copy of the arena uses the same element offsets as byte offsets)
  auto putw = [&](std::vector<ParamInfo>& v, int i) { if (i >= 0) { v[i].warena_off = w; w += (v[i].numel + 15) & ~15L; } };
  auto t5 = [&](std::vector<ParamInfo>& v, T5Idx& ix, bool is_main) {
    putw(v, ix.shared);  // tied LM head operand (and nothing else: embeddings are gathered from the f32 master)
    for (auto& l : ix.dec) { putw(v, l.q); putw(v, l.k); putw(v, l.v); putw(v, l.o); putw(v, l.cq); putw(v, l.co); putw(v, l.wi); putw(v, l.wo); }
    if (is_main && !ix.dec.empty()) {
      e->kvall_w_off = w;
      for (auto& l : ix.dec) { putw(v, l.ck); putw(v, l.cv); }
    }
    for (auto& l : ix.enc) { putw(v, l.q); putw(v, l.k); putw(v, l.v); putw(v, l.o); putw(v, l.wi); putw(v, l.wo); }
  };
  t5(e->P[2], e->mi, true);
  {  // lang: encoder only; shared is gathered in f32 => no arena copy
    auto& v = e->P[1];
    for (auto& l : e->li.enc) { putw(v, l.q); putw(v, l.k); putw(v, l.v); putw(v, l.o); putw(v, l.wi); putw(v, l.wo); }
  }
  {
    auto& v = e->P[0];
    {  // patch-embedding weight [C0, K0]: bf16 rows are stored at a pitch of K0 rounded up to 32 with zero padding, so that
       // the LDS-DMA GEMM (K % 32 == 0) never reads past a row
      const klab_swin_cfg& sc = e->cfg.swin;
      e->pe_k0 = sc.in_ch * sc.patch * sc.patch;
      e->pe_kp = (e->cfg.dtype == KLAB_BF16 && sc.patch == 4) ? ((e->pe_k0 + 31) & ~31) : e->pe_k0;
      v[e->si.pew].warena_off = w;
      w += ((long)sc.embed_dim * e->pe_kp + 15) & ~15L;
    }
    for (auto& st : e->si.st) {
      for (auto& k : st.blk) {
        putw(v, k.qw); putw(v, k.kw); putw(v, k.vw); putw(v, k.pw); putw(v, k.f1w); putw(v, k.f2w);
        // fused q|k|v bias vector [3C] in the f32 side arena (k has no bias: its slot stays zero)
        const long C = v[k.qw].shape[0];
        if (k.qb >= 0) { v[k.qb].farena_off = f; v[k.vb].farena_off = f + 2 * C; }
        f += 3 * C;
      }
      putw(v, st.redw);
    }
  }
  e->warena_elems = w;
  e->farena_elems = f > 0 ? f : 8;
}

// flat gradient layouts.  main: segment 0 = [small decoder params | shared | decoder weights | cross k|v of
// all layers], segment 1 = [small encoder params | encoder weights].  swin: segment 2 = [small | weights].
void plan_grads(klab_engine* e) {
  {
    auto& v = e->P[2];
    long g = 0;
    auto put = [&](int i) { if (i >= 0) { v[i].grad_off = g; g += pad8(v[i].numel); } };
    e->seg_off[0] = 0; e->seg_zero_off[0] = 0;
    for (auto& l : e->mi.dec) { put(l.ln0); put(l.ln1); put(l.ln2); put(l.relb); }
    put(e->mi.dec_final);
    e->seg_zero_len[0] = g;
    put(e->mi.shared);
    e->buckets[0].assign(e->mi.dec.size(), klab_engine::Bucket{0, 0});
    for (size_t li = 0; li < e->mi.dec.size(); ++li) {
      auto& l = e->mi.dec[li];
      const long g0 = g;
      put(l.q); put(l.k); put(l.v); put(l.o); put(l.cq); put(l.co); put(l.wi); put(l.wo);
      e->buckets[0][e->mi.dec.size() - 1 - li] = klab_engine::Bucket{g0, g - g0};  // backward visits the last layer first
    }
    e->kvall_g_off = g;
    for (auto& l : e->mi.dec) { put(l.ck); put(l.cv); }
    e->seg_len[0] = g;
    e->seg_off[1] = g; e->seg_zero_off[1] = g;
    for (auto& l : e->mi.enc) { put(l.ln0); put(l.ln2); put(l.relb); }
    put(e->mi.enc_final);
    e->seg_zero_len[1] = g - e->seg_off[1];
    e->buckets[1].assign(e->mi.enc.size(), klab_engine::Bucket{0, 0});
    for (size_t li = 0; li < e->mi.enc.size(); ++li) {
      auto& l = e->mi.enc[li];
      const long g0 = g;
      put(l.q); put(l.k); put(l.v); put(l.o); put(l.wi); put(l.wo);
      e->buckets[1][e->mi.enc.size() - 1 - li] = klab_engine::Bucket{g0, g - g0};
    }
    e->seg_len[1] = g - e->seg_off[1];
    e->grad_elems[2] = g;
  }
  if (e->cfg.train_swin) {
    auto& v = e->P[0];
    long g = 0;
    auto put = [&](int i) { if (i >= 0) { v[i].grad_off = g; g += pad8(v[i].numel); } };
    put(e->si.peb); put(e->si.penw); put(e->si.penb); put(e->si.lnw); put(e->si.lnb);
    for (auto& st : e->si.st) {
      for (auto& k : st.blk) {
        put(k.ls); put(k.c0w); put(k.c0b); put(k.c2w); put(k.qb); put(k.vb); put(k.pb); put(k.ln1w); put(k.ln1b);
        put(k.f1b); put(k.f2b); put(k.ln2w); put(k.ln2b);
      }
      put(st.mnw); put(st.mnb);
    }
    e->seg_zero_off[2] = 0; e->seg_zero_len[2] = g;
    put(e->si.pew);
    std::vector<klab_engine::Bucket> fwd_order;
    for (auto& st : e->si.st) {
      for (auto& k : st.blk) {
        const long g0 = g;
        put(k.qw); put(k.kw); put(k.vw); put(k.pw); put(k.f1w); put(k.f2w);
        fwd_order.push_back(klab_engine::Bucket{g0, g - g0});
      }
      put(st.redw);
    }
    e->buckets[2].assign(fwd_order.rbegin(), fwd_order.rend());
    e->seg_off[2] = 0; e->seg_len[2] = g;
    e->grad_elems[0] = g;
  }
}

inline uint32_t tag_of(int stack, int layer, int site) { return ((uint32_t)stack << 16) | ((uint32_t)layer << 4) | (uint32_t)site; }
enum { SITE_IN = 0, SITE_PROB = 1, SITE_ATTN_OUT = 2, SITE_XPROB = 3, SITE_XOUT = 4, SITE_MID = 5, SITE_FFN_OUT = 6, SITE_FINAL = 7 };
enum { STACK_LANG = 0, STACK_ENC = 1, STACK_DEC = 2 };

// ------------------------------------------------------------------------------------------------
// workspace plan
// ------------------------------------------------------------------------------------------------
void plan_t5_stack(Bump& b, const klab_t5_cfg& c, int nl, bool dec, bool save, int M, int B, int L, size_t es, T5StackBufs& s,
                   int Lkv) {
  const long d = c.d_model, inner = (long)c.n_heads * c.d_kv, ff = c.d_ff, H = c.n_heads;
  s.M = M; s.Lseq = L;
  const int nsub = dec ? 3 : 2;
  const int nh = save ? nsub * nl + 1 : 2;
  s.h.resize(nsub * nl + 1);
  std::vector<float*> pool(nh);
  for (int i = 0; i < nh; ++i) pool[i] = (float*)b.take((size_t)M * d * 4);
  for (int i = 0; i <= nsub * nl; ++i) s.h[i] = save ? pool[i] : pool[i & 1];
  s.L.resize(nl);
  T5LayerBufs shared_l{};
  for (int i = 0; i < nl; ++i) {
    T5LayerBufs l{};
    if (save || i == 0) {
      l.xn1 = b.take((size_t)M * d * es); l.qkv = b.take((size_t)M * 3 * inner * es); l.ctx = b.take((size_t)M * inner * es);
      l.lse = (float*)b.take((size_t)B * H * L * 4); l.rstd1 = (float*)b.take((size_t)M * 4);
      if (dec) {
        l.xn2 = b.take((size_t)M * d * es); l.qc = b.take((size_t)M * inner * es); l.ctx2 = b.take((size_t)M * inner * es);
        l.lse2 = (float*)b.take((size_t)B * H * L * 4); l.rstd2 = (float*)b.take((size_t)M * 4);
      }
      l.xn3 = b.take((size_t)M * d * es); l.hmid = b.take((size_t)M * ff * es); l.rstd3 = (float*)b.take((size_t)M * 4);
      shared_l = l;
    } else {
      l = shared_l;
    }
    s.L[i] = l;
  }
  s.out_t = b.take((size_t)M * d * es);
  s.rstd_f = (float*)b.take((size_t)M * 4);
  s.bias = (float*)b.take((size_t)H * L * L * 4);
  s.dbias = save ? (float*)b.take((size_t)H * L * L * 4) : nullptr;
  (void)Lkv;
}

size_t plan_workspace(klab_engine* e, void* base, int B, int Ls, int Lt) {
  Bump b(base);
  const klab_model_cfg& c = e->cfg;
  const size_t es = c.dtype == KLAB_BF16 ? 2 : 4;
  e->es = es;
  const int R0 = c.swin.image_size / c.swin.patch;
  const int Rl = R0 >> (c.swin.n_stages - 1);
  const int Nimg = Rl * Rl, Le = Nimg + Ls;
  e->B = B; e->Ls = Ls; e->Lt = Lt; e->Le = Le; e->Nimg = Nimg;
  const long d = c.main.d_model, inner = (long)c.main.n_heads * c.main.d_kv, ff = c.main.d_ff;
  const long Me = (long)B * Le, Md = (long)B * Lt;
  const int nld = c.main.n_dec_layers;

  e->seed_dev = (uint32_t*)b.take(512);   // [0] seed of the current step, [1] base, [2] step counter
  e->err_dev = (int*)((char*)e->seed_dev + 64);
  e->inv_n = (float*)((char*)e->seed_dev + 128);
  e->loss = (float*)((char*)e->seed_dev + 192);
  e->dloss_buf = (float*)((char*)e->seed_dev + 256);
  e->pixels_buf = (float*)b.take((size_t)B * c.swin.in_ch * c.swin.image_size * c.swin.image_size * 4);
  e->src_buf = (long long*)b.take((size_t)B * Ls * 8);
  e->tgt_buf = (long long*)b.take((size_t)B * Lt * 8);
  e->warena = b.take((size_t)e->warena_elems * es);
  e->farena = (float*)b.take((size_t)e->farena_elems * 4);
  if (e->fp8) {
    e->w8 = b.take((size_t)e->warena_elems);
    e->wscale = (float*)b.take((size_t)(e->warena_elems / 8 + 1) * 4);
    e->qdesc = b.take(sizeof(long) * 4 * (e->P[0].size() + e->P[1].size() + e->P[2].size() + 1));
    // activation staging: the widest A operand of any forward Linear (rows x K bytes) and its row scales
    const long R0q = c.swin.image_size / c.swin.patch;
    long mk = (long)B * (Le > Lt ? Le : Lt) * (c.main.d_ff > c.main.d_model ? c.main.d_ff : c.main.d_model);
    long rows = (long)B * (Le > Lt ? Le : Lt);
    for (int st = 0; st < c.swin.n_stages; ++st) {
      const long M = (long)B * (R0q >> st) * (R0q >> st), C = (long)c.swin.embed_dim << st;
      if (M * C * c.swin.mlp_ratio > mk) mk = M * C * c.swin.mlp_ratio;
      if (M > rows) rows = M;
    }
    e->x8_bytes = mk; e->xscale_rows = rows;
    e->x8 = b.take((size_t)mk);
    e->xscale = (float*)b.take((size_t)rows * 4);
    e->x8s_bytes = (long)B * Ls * (c.lang.d_ff > c.lang.d_model ? c.lang.d_ff : c.lang.d_model); e->xscales_rows = (long)B * Ls;
    e->x8s = b.take((size_t)e->x8s_bytes);
    e->xscales = (float*)b.take((size_t)e->xscales_rows * 4);
  }
  // two descriptor groups (trainable / frozen); the patch-embedding weight contributes one descriptor per row
  e->cast_desc = b.take(sizeof(long) * 3 * 2 * (e->P[0].size() + e->P[1].size() + e->P[2].size() + c.swin.embed_dim + 1));
  e->fcast_desc = b.take(sizeof(long) * 3 * (e->P[0].size() + 1));
  e->adam_desc = b.take(sizeof(long) * 4 * (e->P[2].size() + 1));
  {
    const int nle = c.main.n_layers, nldx = c.main.n_dec_layers;
    const int ncmax = (2 * nle + 1) > (3 * nldx + 1) ? (2 * nle + 1) : (3 * nldx + 1);
    const long Mmax = (long)B * (Le > Lt ? Le : Lt);
    e->rms_part_stride = (long)klab_rmsnorm_part_rows((int)Mmax) * c.main.d_model;
    e->rms_part = (float*)b.take((size_t)ncmax * e->rms_part_stride * 4);
    e->rms_dst_dev[0] = (float**)b.take(sizeof(float*) * (2 * nle + 1));
    e->rms_dst_dev[1] = (float**)b.take(sizeof(float*) * (3 * nldx + 1));
  }

  plan_t5_stack(b, c.lang, c.lang.n_layers, false, false, B * Ls, B, Ls, es, e->lang, 0);
  plan_t5_stack(b, c.main, c.main.n_layers, false, true, (int)Me, B, Le, es, e->enc, 0);
  plan_t5_stack(b, c.main, nld, true, true, (int)Md, B, Lt, es, e->dec, Le);
  e->kv_all = b.take((size_t)Me * nld * 2 * inner * es);
  e->dkv_all = b.take((size_t)Me * nld * 2 * inner * es);
  e->logits = b.take((size_t)Md * c.main.vocab * es);
  e->dc_h[0] = (float*)b.take((size_t)B * d * 4); e->dc_h[1] = (float*)b.take((size_t)B * d * 4);
  e->dc_rstd = (float*)b.take((size_t)B * 4);
  e->dc_xn = b.take((size_t)B * d * es); e->dc_q = b.take((size_t)B * inner * es); e->dc_ctx = b.take((size_t)B * inner * es);
  e->dc_hmid = b.take((size_t)B * ff * es); e->dc_out = b.take((size_t)B * d * es);
  e->dc_logits = b.take((size_t)B * c.main.vocab * es);
  e->loss_row = (float*)b.take((size_t)Md * 4);
  const long Mx = Me > Md ? Me : Md;
  e->dh_a = (float*)b.take((size_t)Mx * d * 4);
  e->dh_b = (float*)b.take((size_t)Mx * d * 4);
  e->dxn = (float*)b.take((size_t)Mx * d * 4);
  e->denc = (float*)b.take((size_t)Me * d * 4);
  {
    const int nl = c.main.n_layers > c.main.n_dec_layers ? c.main.n_layers : c.main.n_dec_layers;
    e->dy_pool.assign(3 * nl + 2, nullptr); e->dhmid_pool.assign(nl, nullptr); e->dqkv_pool.assign(nl, nullptr);
    e->dqc_pool.assign(c.main.n_dec_layers, nullptr);
    for (auto& q : e->dy_pool) q = b.take((size_t)Mx * d * es);
    for (auto& q : e->dhmid_pool) q = b.take((size_t)Mx * ff * es);
    for (auto& q : e->dqkv_pool) q = b.take((size_t)Mx * 3 * inner * es);
    for (auto& q : e->dqc_pool) q = b.take((size_t)Md * inner * es);
  }
  e->dctx = b.take((size_t)Mx * inner * es);
  {
    const long Lmax = Le > Lt ? Le : Lt;
    const int nlmax = c.main.n_layers > c.main.n_dec_layers ? c.main.n_layers : c.main.n_dec_layers;
    e->ds_ws = b.take((size_t)nlmax * B * c.main.n_heads * Lmax * ((Lmax + 31) & ~31L) * es);  // one slab per layer
  }

  // ---- Swin ----
  const klab_swin_cfg& s = c.swin;
  const long T0 = (long)R0 * R0, M0 = (long)B * T0, K0 = (long)s.in_ch * s.patch * s.patch, C0 = s.embed_dim;
  e->cols = b.take((size_t)M0 * ((K0 + 31) & ~31L) * es);  // rows padded to a multiple of 32 columns (bf16 LDS-DMA GEMM path)
  e->pe_out = b.take((size_t)M0 * C0 * es);
  e->pe_mean = (float*)b.take((size_t)M0 * 4); e->pe_rstd = (float*)b.take((size_t)M0 * 4);
  e->x0 = (float*)b.take((size_t)M0 * C0 * 4); e->x0t = b.take((size_t)M0 * C0 * es);
  e->sw.assign(s.n_stages, SwinStageBufs());
  long maxMC = M0 * C0;
  for (int st = 0; st < s.n_stages; ++st) {
    const int R = R0 >> st;
    const long C = C0 << st, M = (long)B * R * R, F = (long)s.mlp_ratio * C;
    const int H = s.heads[st];
    const int w = R < s.window ? R : s.window;
    const int nWr = (R + (w > 0 ? w : 1) - 1) / (w > 0 ? w : 1);  // padded windows count in full (HF/swinv2:645-650)
    const int n = w * w, nW = nWr * nWr;
    const int ntab = (2 * w - 1) * (2 * w - 1);
    if (M * C > maxMC) maxMC = M * C;
    SwinStageBufs& sb = e->sw[st];
    sb.blk.resize(s.depths[st]);
    for (int k = 0; k < s.depths[st]; ++k) {
      SwinBlockBufs& q = sb.blk[k];
      q.R = R; q.w = w; q.H = H; q.C = (int)C; q.M = M;
      q.shift = (k % 2 == 0 || R <= w) ? 0 : s.window / 2;
      q.qkv = b.take((size_t)M * 3 * C * es); q.ctx = b.take((size_t)M * C * es); q.po = b.take((size_t)M * C * es);
      q.lse = (float*)b.take((size_t)B * nW * H * n * 4);
      q.mean1 = (float*)b.take((size_t)M * 4); q.rstd1 = (float*)b.take((size_t)M * 4);
      q.h1 = (float*)b.take((size_t)M * C * 4); q.h1t = b.take((size_t)M * C * es);
      q.z = c.train_swin ? b.take((size_t)M * F * es) : nullptr;
      q.a = b.take((size_t)M * F * es); q.fo = b.take((size_t)M * C * es);
      q.mean2 = (float*)b.take((size_t)M * 4); q.rstd2 = (float*)b.take((size_t)M * 4);
      q.h2 = (float*)b.take((size_t)M * C * 4); q.h2t = b.take((size_t)M * C * es);
      const bool big = n > 64 || (R % w) != 0;  // tiled / streaming attention kernels with the bias as a table (attn_swin_large.hip)
      q.bias = big ? nullptr : (float*)b.take((size_t)H * n * n * 4);
      q.btab = big ? (float*)b.take((size_t)ntab * H * 4) : nullptr;
      q.table = (float*)b.take((size_t)ntab * H * 4);
      q.hidden = (float*)b.take((size_t)ntab * 512 * 4);
    }
    sb.mg = sb.mo = nullptr; sb.mmean = sb.mrstd = nullptr; sb.xm = nullptr; sb.xmt = nullptr;
    if (st < s.n_stages - 1) {
      const long M2 = M / 4;
      sb.mg = b.take((size_t)M2 * 4 * C * es); sb.mo = b.take((size_t)M2 * 2 * C * es);
      sb.mmean = (float*)b.take((size_t)M2 * 4); sb.mrstd = (float*)b.take((size_t)M2 * 4);
      sb.xm = (float*)b.take((size_t)M2 * 2 * C * 4); sb.xmt = b.take((size_t)M2 * 2 * C * es);
    }
  }
  const long Ml = (long)B * Nimg;
  e->sw_fmean = (float*)b.take((size_t)Ml * 4); e->sw_frstd = (float*)b.take((size_t)Ml * 4);
  if (c.train_swin) {
    const long F0 = (long)s.mlp_ratio;
    e->sdh_a = (float*)b.take((size_t)maxMC * 4); e->sdh_b = (float*)b.take((size_t)maxMC * 4);
    e->sdm = (float*)b.take((size_t)maxMC * 4);
    e->sdy = b.take((size_t)maxMC * es); e->sdctx = b.take((size_t)maxMC * es);
    e->sdqkv = b.take((size_t)maxMC * 3 * es); e->sda = b.take((size_t)maxMC * F0 * es);
    for (int pz = 0; pz < 2; ++pz) {
      e->sdyA[pz] = b.take((size_t)maxMC * es); e->sdyB[pz] = b.take((size_t)maxMC * es);
      e->sdaP[pz] = b.take((size_t)maxMC * F0 * es); e->sdqkvP[pz] = b.take((size_t)maxMC * 3 * es);
    }
    int maxn = 0, maxH = 0, maxn_small = 1;
    for (int st = 0; st < s.n_stages; ++st) {
      const int R = R0 >> st; const int w = R < s.window ? R : s.window;
      if (w * w > maxn) maxn = w * w;
      if (w * w <= 64 && w * w > maxn_small) maxn_small = w * w;
      if (s.heads[st] > maxH) maxH = s.heads[st];
    }
    int nblk = 0;
    for (int st = 0; st < s.n_stages; ++st) nblk += s.depths[st];
    e->sdb_stride = ((long)maxH * maxn_small * maxn_small + 63) & ~63L;  // dense d(bias) of the one-tile windows, floats per block
    e->sdt_stride = ((long)(4 * maxn) * maxH + 63) & ~63L;               // d(bias table) of the large windows / d(table), floats per block
    e->sdz_bytes = (size_t)nblk * (e->sdb_stride + 2 * e->sdt_stride) * 4;
    e->sdbias = (float*)b.take(e->sdz_bytes);
    e->sdbtab = e->sdbias + (long)nblk * e->sdb_stride;
    e->sdtable = e->sdbtab + (long)nblk * e->sdt_stride;
  }
  {  // scratch of the matrix-core window attention: backward of the one-tile windows (train_swin), and forward + backward of
     // windows of more than 64 tokens (window-major copies for the streaming kernels)
    e->sattn_ws = nullptr; e->sattn_ws_bytes = 0;
    const char* ev = getenv("KLAB_SWIN_BWD_MFMA");  // "0": keep the vector-ALU window-attention kernels (A/B switch)
    if (!(ev && ev[0] == '0')) {
      size_t need = 0;
      for (int st = 0; st < s.n_stages; ++st) {
        const int R = R0 >> st; const int w = R < s.window ? R : s.window;
        if (!c.train_swin && w * w <= 64 && R % w == 0) continue;  // frozen tower, one-tile unpadded window: the forward kernels need no scratch
        const size_t x = klab_swin_attn_bwd_ws_bytes(c.dtype, B, R, w, s.heads[st], s.embed_dim << st);
        if (x > need) need = x;  // (0: that stage is outside the matrix-core envelope and runs its own kernels)
      }
      if (need) { e->sattn_ws = b.take(need); e->sattn_ws_bytes = need; }
    }
  }
  return (b.off + 255) & ~(size_t)255;
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
struct Ctx {
  klab_engine* e; hipStream_t s; int dt; size_t es;
  void* ws() const { return (void*)s; }
};

klab_gemm_args G0(const Ctx& c, int M, int N, int K, const void* A, long lda, int ak, const void* B, long ldb, int bk, void* C,
                  long ldc, int cdt) {
  klab_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.M = M; g.N = N; g.K = K; g.dtype = c.dt;
  g.A = A; g.lda = lda; g.a_kmajor = ak; g.B = B; g.ldb = ldb; g.b_kmajor = bk;
  g.C = C; g.ldc = ldc; g.c_dtype = cdt; g.alpha = 1.f;
  return g;
}
inline void* woff(const Ctx& c, long off) { return (char*)c.e->warena + (size_t)off * c.es; }
inline void* eoff(const Ctx& c, void* p, long elems) { return (char*)p + (size_t)elems * c.es; }

inline bool fp8_weight_ok(long K) { return (K & 15) == 0 && K >= 16 && K <= 4096; }
// A forward Linear product x[M,K] @ W[N,K]^T (+ epilogue).  fp8 mode: x is quantised per token into the stream's staging
// buffer and the product runs on the fp8 matrix cores against the e4m3 copy of W (arena offset woffv); otherwise klab_gemm.
int fwd_gemm(const Ctx& c, klab_gemm_args& g, long woffv) {
  klab_engine* e = c.e;
  if (e->fp8 && woffv >= 0 && !(woffv & 15) && g.a_kmajor && g.b_kmajor && !g.accumulate && fp8_weight_ok(g.K) && g.lda == g.K && g.ldb == g.K) {
    const bool side = c.s == e->side;
    void* x8 = side ? e->x8s : e->x8;
    float* xs = side ? e->xscales : e->xscale;
    const long cap = side ? e->x8s_bytes : e->x8_bytes, rows = side ? e->xscales_rows : e->xscale_rows;
    if ((long)g.M * g.K <= cap && g.M <= rows) {
      const bool have = e->q8_src[side] == g.A && e->q8_rows[side] == g.M && e->q8_k[side] == g.K;  // the norm in front of this Linear already quantised its rows
      e->q8_src[side] = nullptr;
      if (!have) RC(klab_quant_fp8_rows(g.A, g.lda, g.M, g.K, x8, g.K, xs, c.ws()));
      klab_gemm_args q = g;
      q.A = x8; q.B = (const char*)e->w8 + woffv;
      return klab_gemm_fp8(&q, xs, e->wscale + woffv / 8, g.K / 8, c.ws());
    }
  }
  return klab_gemm(&g, c.ws());
}

// T5 RMS-norm in front of a Linear: in fp8 mode the kernel also leaves the rows in e4m3 in the stream's staging buffer
int rms_fwd_for_linear(const Ctx& c, const float* x, const float* w, void* y, float* rstd, int M, int d, float eps) {
  klab_engine* e = c.e;
  static const bool fuse = [] { const char* v = getenv("KLAB_FP8_NORM_QUANT"); return !v || atoi(v) != 0; }();
  if (e->fp8 && fuse && c.dt == KLAB_BF16 && d <= 1024 && fp8_weight_ok(d)) {
    const bool side = c.s == e->side;
    const long cap = side ? e->x8s_bytes : e->x8_bytes, rows = side ? e->xscales_rows : e->xscale_rows;
    if ((long)M * d <= cap && M <= rows) {
      RC(klab_rmsnorm_fwd_q8(x, w, y, rstd, side ? e->x8s : e->x8, side ? e->xscales : e->xscale, M, d, eps, 0.f, nullptr, 0, c.ws()));
      e->q8_src[side] = y; e->q8_rows[side] = M; e->q8_k[side] = d;
      return 0;
    }
  }
  return klab_rmsnorm_fwd(x, w, y, c.dt, nullptr, rstd, M, d, eps, 0, 0, 0, 0.f, nullptr, 0, c.ws());
}

// Swin LayerNorm (+ shortcut) / GELU in front of a Linear: the fp8-mode forms that leave the rows in e4m3 as well
static bool q8_room(const Ctx& c, long M, long K, bool& side) {
  klab_engine* e = c.e;
  static const bool fuse = [] { const char* v = getenv("KLAB_FP8_NORM_QUANT"); return !v || atoi(v) != 0; }();
  side = c.s == e->side;
  if (!e->fp8 || !fuse || c.dt != KLAB_BF16 || !fp8_weight_ok(K)) return false;
  return M * K <= (side ? e->x8s_bytes : e->x8_bytes) && M <= (side ? e->xscales_rows : e->xscale_rows);
}
int ln_fwd_for_linear(const Ctx& c, const void* y, const float* g, const float* b, const float* shortcut, float* out, void* outt, float* mean,
                      float* rstd, int M, int C, float eps) {
  klab_engine* e = c.e;
  bool side;
  if (q8_room(c, M, C, side)) {
    RC(klab_layernorm_fwd_q8(y, g, b, shortcut, out, outt, mean, rstd, side ? e->x8s : e->x8, side ? e->xscales : e->xscale, M, C, eps, c.ws()));
    e->q8_src[side] = outt; e->q8_rows[side] = M; e->q8_k[side] = C;
    return 0;
  }
  return klab_layernorm_fwd(y, c.dt, g, b, shortcut, out, outt, c.dt, mean, rstd, M, C, eps, 0, 0, 0, 0.f, nullptr, 0, c.ws());
}
int gelu_fwd_for_linear(const Ctx& c, const void* z, void* a, int M, int F) {
  klab_engine* e = c.e;
  bool side;
  if (F <= 4096 && !(F & 7) && q8_room(c, M, F, side)) {
    RC(klab_gelu_fwd_q8(z, a, side ? e->x8s : e->x8, side ? e->xscales : e->xscale, M, F, c.ws()));
    e->q8_src[side] = a; e->q8_rows[side] = M; e->q8_k[side] = F;
    return 0;
  }
  return klab_gelu_fwd(z, a, c.dt, (long)M * F, c.ws());
}

// y = x @ W^T   (W [N,K] from the weight arena)
int linear_fwd(const Ctx& c, const void* x, int M, int K, long woffv, int N, void* y, long ldy, int ydt, const float* bias = nullptr,
               int act = 0) {
  klab_gemm_args g = G0(c, M, N, K, x, K, 1, woff(c, woffv), K, 1, y, ldy, ydt);
  g.bias = bias; g.act = act;
  return fwd_gemm(c, g, woffv);
}
// dX[M,K] = dY[M,N] @ W[N,K]
int linear_dgrad(const Ctx& c, const void* dy, long lddy, int M, int N, long woffv, int K, void* dx, int dxdt) {
  klab_gemm_args g = G0(c, M, K, N, dy, lddy, 1, woff(c, woffv), K, 0, dx, K, dxdt);
  return klab_gemm(&g, c.ws());
}
// dW[N,K] = dY[M,N]^T @ X[M,K]  -> f32 grads
int linear_wgrad(const Ctx& c, const void* dy, long lddy, const void* x, long ldx, int M, int N, int K, float* dw) {
  klab_gemm_args g = G0(c, N, K, M, dy, lddy, 0, x, ldx, 0, dw, K, KLAB_F32);
  g.accumulate = 1; g.atomic_ok = 1;  // the segment's grad slice was zeroed: small outputs may split K
  return klab_gemm(&g, c.ws());
}

int t5_sublayer_out(const Ctx& c, const void* x, int M, int K, long woffv, int d, const float* resid, float* hout, float p, uint32_t tag) {
  klab_gemm_args g = G0(c, M, d, K, x, K, 1, woff(c, woffv), K, 1, hout, d, KLAB_F32);
  g.residual = resid; g.ldr = d; g.r_dtype = KLAB_F32;
  g.drop_p = p; g.seed_dev = c.e->seed_dev; g.drop_tag = tag;
  return fwd_gemm(c, g, woffv);
}

// A/B switch of the fused attention front half (klab_t5_attn_fused_fwd): KLAB_T5_ATTN_FUSED=0 keeps the three launches
static bool attn_fused_on() {
  static const bool on = [] { const char* v = getenv("KLAB_T5_ATTN_FUSED"); return !v || atoi(v) != 0; }();
  return on;
}

// ------------------------------------------------------------------------------------------------
// T5 stack forward (HF/t5:663-750); `h[0]` must already hold dropout(inputs_embeds)
// ------------------------------------------------------------------------------------------------
int t5_stack_forward(const Ctx& c, const klab_t5_cfg& cfg, const std::vector<ParamInfo>& P, const std::vector<const float*>& W,
                     const std::vector<T5LayerIdx>& L, int final_ln, T5StackBufs& s, bool dec, int stack_id, float p, int B,
                     const void* kv_all, int Lkv, long kv_ld, float* out_f32, int grp, int grp_stride, int off, float p_final) {
  const int d = cfg.d_model, H = cfg.n_heads, dk = cfg.d_kv, inner = H * dk, ff = cfg.d_ff;
  const int M = s.M, Lq = s.Lseq;
  const int relb = L[0].relb;
  if (c.e->bias_ready[stack_id]) c.e->bias_ready[stack_id] = false;  // computed ahead, beside the Swin tower (forward_part_a)
  else RC(klab_relbias_fwd(W[relb], s.bucket, s.bias, H, Lq, Lq, c.ws()));
  int j = 0;
  for (size_t i = 0; i < L.size(); ++i) {
    const T5LayerIdx& l = L[i];
    T5LayerBufs& b = s.L[i];
    // --- self attention (HF/t5:372-401) ---
    {
      klab_attn_args a;
      memset(&a, 0, sizeof(a));
      a.dtype = c.dt; a.q = b.qkv; a.ldq = 3 * inner; a.k = eoff(c, b.qkv, inner); a.ldk = 3 * inner;
      a.v = eoff(c, b.qkv, 2 * inner); a.ldv = 3 * inner; a.bias = s.bias; a.causal = dec ? 1 : 0;
      a.ctx = b.ctx; a.ldo = inner; a.lse = b.lse; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lq; a.dk = dk;
      a.drop_p = p; a.seed_dev = c.e->seed_dev; a.drop_tag = tag_of(stack_id, (int)i, SITE_PROB);
      // one launch for norm -> q|k|v -> attention where the fused kernel's envelope allows (bf16, d_model 512, head dim 64, <= 64
      // tokens per sample: T5-small at the caption shapes); otherwise the three launches
      int frc = KLAB_ERR_UNSUPPORTED;
      if (attn_fused_on() && c.dt == KLAB_BF16 && !c.e->fp8 && Lq >= 32) {  // (the 9-token language encoder: one row tile per sample would stream the weights for nothing)
        klab_attn_fused_args fa;
        memset(&fa, 0, sizeof(fa));
        fa.x = s.h[j]; fa.gamma = W[l.ln0]; fa.eps = cfg.ln_eps; fa.d_model = d; fa.w = woff(c, P[l.q].warena_off);
        fa.xn = b.xn1; fa.rstd = b.rstd1; fa.proj = b.qkv; fa.ldproj = 3 * inner; fa.cross = 0; fa.attn = a;
        frc = klab_t5_attn_fused_fwd(&fa, c.ws());
        if (frc != 0 && frc != KLAB_ERR_UNSUPPORTED) return frc;
      }
      if (frc != 0) {
        RC(rms_fwd_for_linear(c, s.h[j], W[l.ln0], b.xn1, b.rstd1, M, d, cfg.ln_eps));
        RC(linear_fwd(c, b.xn1, M, d, P[l.q].warena_off, 3 * inner, b.qkv, 3 * inner, c.dt));
        RC(klab_t5_attn_fwd(&a, c.ws()));
      }
    }
    RC(t5_sublayer_out(c, b.ctx, M, inner, P[l.o].warena_off, d, s.h[j], s.h[j + 1], p, tag_of(stack_id, (int)i, SITE_ATTN_OUT)));
    ++j;
    if (dec) {  // --- cross attention (HF/t5:404-432), K/V of all layers projected once ---
      klab_attn_args a;
      memset(&a, 0, sizeof(a));
      a.dtype = c.dt; a.q = b.qc; a.ldq = inner;
      a.k = eoff(c, (void*)kv_all, (long)i * 2 * inner); a.ldk = kv_ld;
      a.v = eoff(c, (void*)kv_all, (long)i * 2 * inner + inner); a.ldv = kv_ld;
      a.bias = nullptr; a.causal = 0; a.ctx = b.ctx2; a.ldo = inner; a.lse = b.lse2;
      a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lkv; a.dk = dk;
      a.drop_p = p; a.seed_dev = c.e->seed_dev; a.drop_tag = tag_of(stack_id, (int)i, SITE_XPROB);
      int frc = KLAB_ERR_UNSUPPORTED;
      if (attn_fused_on() && c.dt == KLAB_BF16 && !c.e->fp8 && Lq >= 32) {  // norm -> q projection -> attention over the projected encoder output
        klab_attn_fused_args fa;
        memset(&fa, 0, sizeof(fa));
        fa.x = s.h[j]; fa.gamma = W[l.ln1]; fa.eps = cfg.ln_eps; fa.d_model = d; fa.w = woff(c, P[l.cq].warena_off);
        fa.xn = b.xn2; fa.rstd = b.rstd2; fa.proj = b.qc; fa.ldproj = inner; fa.cross = 1; fa.attn = a;
        frc = klab_t5_attn_fused_fwd(&fa, c.ws());
        if (frc != 0 && frc != KLAB_ERR_UNSUPPORTED) return frc;
      }
      if (frc != 0) {
        RC(rms_fwd_for_linear(c, s.h[j], W[l.ln1], b.xn2, b.rstd2, M, d, cfg.ln_eps));
        RC(linear_fwd(c, b.xn2, M, d, P[l.cq].warena_off, inner, b.qc, inner, c.dt));
        RC(klab_t5_attn_fwd(&a, c.ws()));
      }
      RC(t5_sublayer_out(c, b.ctx2, M, inner, P[l.co].warena_off, d, s.h[j], s.h[j + 1], p, tag_of(stack_id, (int)i, SITE_XOUT)));
      ++j;
    }
    // --- feed forward (HF/t5:83-94,137-141) ---
    RC(rms_fwd_for_linear(c, s.h[j], W[l.ln2], b.xn3, b.rstd3, M, d, cfg.ln_eps));
    {
      klab_gemm_args g = G0(c, M, ff, d, b.xn3, d, 1, woff(c, P[l.wi].warena_off), d, 1, b.hmid, ff, c.dt);
      g.act = KLAB_ACT_RELU; g.drop_p = p; g.seed_dev = c.e->seed_dev; g.drop_tag = tag_of(stack_id, (int)i, SITE_MID);
      RC(fwd_gemm(c, g, P[l.wi].warena_off));
    }
    RC(t5_sublayer_out(c, b.hmid, M, ff, P[l.wo].warena_off, d, s.h[j], s.h[j + 1], p, tag_of(stack_id, (int)i, SITE_FFN_OUT)));
    ++j;
  }
  // final norm + dropout (HF/t5:744-745)
  RC(klab_rmsnorm_fwd(s.h[j], W[final_ln], out_f32 ? nullptr : s.out_t, c.dt, out_f32, s.rstd_f, M, d, cfg.ln_eps, grp, grp_stride, off,
                      p_final, c.e->seed_dev, out_f32 ? tag_of(STACK_ENC, 0, SITE_IN) : tag_of(stack_id, 0, SITE_FINAL), c.ws()));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// side-stream helpers: "everything enqueued on the main stream so far" -> visible to the side stream, and back
// ------------------------------------------------------------------------------------------------
int side_after_main(const Ctx& c) {
  klab_engine* e = c.e;
  hipEvent_t ev = e->evpool[e->ev_next++ % e->evpool.size()];
  RC((int)hipEventRecord(ev, c.s));
  return (int)hipStreamWaitEvent(e->side, ev, 0);
}
int main_after_side(const Ctx& c) {
  klab_engine* e = c.e;
  hipEvent_t ev = e->evpool[e->ev_next++ % e->evpool.size()];
  RC((int)hipEventRecord(ev, e->side));
  return (int)hipStreamWaitEvent(c.s, ev, 0);
}
// Weight gradients on the side stream.  Their operands live in per-sub-layer buffers that are not rewritten inside a
// backward segment, so they may run any time after the producer: they are queued here and released in batches with ONE
// main->side event per layer (an event record costs the main queue ~6 us; one per weight gradient left ~40 us of
// bubbles in every layer's critical path).
struct PendingWgrad { const void* dy; long lddy; const void* x; long ldx; int M, N, K; float* dw; };
struct WgradQueue {
  std::vector<PendingWgrad> q;
  int layers = 0;  // flush calls (= layers) the queue spans
  void push(const void* dy, long lddy, const void* x, long ldx, int M, int N, int K, float* dw) { q.push_back({dy, lddy, x, ldx, M, N, K, dw}); }
  // Called once per layer.  now == false: keep the layer's products queued (the caller's plan releases them with a later layer's);
  // now == true: release everything queued in ONE grouped launch -- on 256 x 256 tiles (mm8p.hip: half the operand bytes of the
  // 128-wide grouping) when the list spans two layers or more, on the 128-wide kernel when it is one layer's.
  int flush(const Ctx& c, bool now = true, bool* launched = nullptr) {
    if (launched) *launched = false;
    ++layers;
    if (q.empty()) { layers = 0; return 0; }
    if (!now && q.size() + 8 <= 32) return 0;
    if (launched) *launched = true;
    const bool large = layers >= 2;  // a list that spans layers goes to the 256 x 256 grouped kernel
    layers = 0;
    // timing diagnostics only (results are wrong / unoverlapped): KLAB_DIAG_WGRAD=skip drops the layer's weight gradients,
    // =main runs them on the main stream behind the layer's chain instead of beside it
    static const int diag = [] { const char* v = getenv("KLAB_DIAG_WGRAD"); return !v ? 0 : (v[0] == 's' ? 1 : (v[0] == 'm' ? 2 : 0)); }();
    if (diag == 1) { q.clear(); return 0; }
    if (diag != 2) RC(side_after_main(c));
    Ctx cs{c.e, diag == 2 ? c.s : c.e->side, c.dt, c.es};
    std::vector<klab_gemm_args> gs;  // one grouped launch for the layer's weight gradients
    gs.reserve(q.size());
    for (const PendingWgrad& w : q) {
      klab_gemm_args g = G0(cs, w.N, w.K, w.M, w.dy, w.lddy, 0, w.x, w.ldx, 0, w.dw, w.K, KLAB_F32);  // as linear_wgrad
      g.accumulate = 1; g.atomic_ok = 1;
      gs.push_back(g);
    }
    klab_engine::Probe& pr = c.e->probe[1];
    const bool probe = c.e->probe_on && pr.n < (int)pr.a.size();
    // (the grouped launch itself carries the events as its start / stop events: no extra packets on the stream)
    if (probe) { klab::tl_launch_probe.a = pr.a[pr.n]; klab::tl_launch_probe.b = pr.b[pr.n]; }
    klab::tl_grouped_large_tiles = large;
    const int grc = klab_gemm_grouped(gs.data(), (int)gs.size(), cs.ws());
    klab::tl_grouped_large_tiles = false;
    RC(grc);
    if (probe) {
      if (klab::tl_launch_probe.a) klab::tl_launch_probe.a = nullptr;  // no grouped launch happened (members went through klab_gemm)
      else {
        double fl = 0;
        for (const PendingWgrad& w : q) fl += 2.0 * w.M * (double)w.N * w.K;
        pr.flops[pr.n++] = fl;
      }
    }
    q.clear();
    return 0;
  }
};

// ------------------------------------------------------------------------------------------------
// T5 stack backward.  In: dxn = d loss / d (final-norm output) in f32 [M,d].  Out: dh_cur = d loss / d h[0].
// The activation-gradient chain (dgrad GEMMs, attention, norms) runs on the main stream; every weight gradient is
// issued to the side stream as soon as its two operands exist, so the two families of small GEMMs overlap.
// ------------------------------------------------------------------------------------------------
int t5_stack_backward(const Ctx& c, const klab_t5_cfg& cfg, const std::vector<ParamInfo>& P, const std::vector<const float*>& W,
                      float* Gflat, const std::vector<T5LayerIdx>& L, int final_ln, T5StackBufs& s, bool dec, int stack_id, float p,
                      int B, const void* kv_all, void* dkv_all, int Lkv, long kv_ld, float** dh_out) {
  klab_engine* e = c.e;
  const int d = cfg.d_model, H = cfg.n_heads, dk = cfg.d_kv, inner = H * dk, ff = cfg.d_ff;
  const int M = s.M, Lq = s.Lseq;
  const int nsub = dec ? 3 : 2;
  int j = nsub * (int)L.size();
  float* dh_cur = e->dh_a;
  float* dh_oth = e->dh_b;
  auto G = [&](int pi) { return Gflat + P[pi].grad_off; };
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  int dy_i = 0, deferred = 0;
  WgradQueue wq;
  // norm-weight gradients: per-workgroup partials now, ONE fixed-order reduction for the whole stack at the end (the
  // 256-way same-address f32 atomics cost ~3 us in each of the stack's 13-19 norm backward kernels)
  const int stk = dec ? 1 : 0;
  const bool part_ok = Gflat == e->G[2] && e->rms_part && d <= 1024 && (long)klab_rmsnorm_part_rows(M) * d <= e->rms_part_stride;
  int rms_calls = 0;
  auto rms_bwd = [&](const float* dyin, const float* x, const float* w, const float* rstd, const float* dres, float* dx, void* dxt, int pidx,
                     float p_y, uint32_t tag_y, float p_prev, uint32_t tag_prev) -> int {
    if (part_ok)
      return klab_rmsnorm_bwd_part(dyin, x, w, rstd, dres, dx, dxt, c.dt, e->rms_part + (long)(rms_calls++) * e->rms_part_stride, M, d, 0, 0, 0,
                                   p_y, tag_y, p_prev, tag_prev, e->seed_dev, c.ws());
    return klab_rmsnorm_bwd(dyin, x, w, rstd, dres, dx, dxt, c.dt, G(pidx), M, d, 0, 0, 0, p_y, tag_y, p_prev, tag_prev, e->seed_dev, c.ws());
  };
  auto next_dy = [&]() { return e->dy_pool[dy_i++ % e->dy_pool.size()]; };
  if (e->dbias_zeroed[stk]) e->dbias_zeroed[stk] = false;  // cleared on the side stream at backward entry
  else RC(hipMemsetAsync(s.dbias, 0, (size_t)H * Lq * Lq * 4, c.s));
  // final norm: y = drop(norm(h[j])); previous sub-layer output dropout = FFN_OUT of the last layer
  void* dy = next_dy();  // masked, compute-dtype gradient of the current sub-layer's GEMM output
  RC(rms_bwd(e->dxn, s.h[j], W[final_ln], s.rstd_f, nullptr, dh_cur, dy, final_ln, p, tag_of(stack_id, 0, SITE_FINAL), p,
             tag_of(stack_id, (int)L.size() - 1, SITE_FFN_OUT)));
  std::vector<int> pending_buckets;  // layers whose weight gradients are queued but not yet launched
  // Release plan of the weight gradients (KLAB_WGRAD_GROUP_TILES = T, default 90; 0 = one 128-wide grouped launch per layer, round
  // 2's form).  One T5-small layer's products are 48-64 tiles of 256 x 256 -- a fifth of the chip, and a large tile takes ~100 us
  // whatever the grid, which is why PER-LAYER large tiles lost (KLAB_WGRAD_P8: 6.41 vs 6.18 ms).  So the layers of a stack except
  // its last are released in groups of >= T tiles (T5-small: decoder 3 + 2 layers, encoder 3 + 2; same box, three rounds, T = 0 / 90
  // / 110 (encoder 5 in one group) / 200: 5.990 / 5.855 / 5.880 / 5.982 ms per step), and the LAST layer alone on the
  // 128-wide kernel: its launch is the tail the segment's end waits for, and 70 us of tail hide behind the main chain's own last
  // kernels where a 180-us group does not.  A layer that reaches T on its own (T5-base / large) keeps the per-layer form.
  const int Ln = (int)L.size();
  std::vector<char> flush_at(Ln, 1);
  static const int group_tiles = [] { const char* v = getenv("KLAB_WGRAD_GROUP_TILES"); return v ? atoi(v) : 90; }();
  static const bool kv_per_layer = [] { const char* v = getenv("KLAB_KV_WGRAD_PER_LAYER"); return !v || atoi(v) != 0; }();
  const bool kv_in_queue = dec && dkv_all && kv_per_layer && Gflat == e->G[2] && !e->use_graph && e->kvall_g_off >= 0;
  if (group_tiles > 0 && Ln >= 3 && Gflat == e->G[2] && !e->use_graph) {
    auto T2 = [](int n, int k) { return (long)((n + 255) / 256) * ((k + 255) / 256); };
    const long tl = T2(d, ff) + T2(ff, d) + T2(d, inner) + T2(3 * inner, d) + (dec ? T2(d, inner) + T2(inner, d) + (kv_in_queue ? T2(2 * inner, d) : 0) : 0);
    const int per_group = (int)((group_tiles + tl - 1) / tl), per_layer = dec ? 7 : 4;
    if (per_group >= 2) {
      const int body = Ln - 1;
      int ng = body / per_group;
      if (ng < 1) ng = 1;
      if (((body + ng - 1) / ng) * per_layer <= 32) {  // (a grouped launch takes 32 products)
        std::fill(flush_at.begin(), flush_at.end(), 0);
        int i = Ln - 1;
        for (int gi = 0; gi < ng; ++gi) { i -= body / ng + (gi < body % ng ? 1 : 0); flush_at[i + 1] = 1; }
        flush_at[0] = 1;
      }
    }
  }
  for (int i = (int)L.size() - 1; i >= 0; --i) {
    const T5LayerIdx& l = L[i];
    T5LayerBufs& b = s.L[i];
    void* dhmid = e->dhmid_pool[i];
    void* dqkv = e->dqkv_pool[i];
    // ---------------- FFN ----------------
    --j;
    wq.push(dy, d, b.hmid, ff, M, d, ff, G(l.wo));
    {
      klab_gemm_args g = G0(c, M, ff, d, dy, d, 1, woff(c, P[l.wo].warena_off), ff, 0, dhmid, ff, c.dt);
      g.aux = b.hmid; g.ldaux = ff; g.aux_mode = KLAB_AUX_NONZERO; g.aux_scale = inv_keep;
      RC(klab_gemm(&g, c.ws()));
    }
    wq.push(dhmid, ff, b.xn3, d, M, ff, d, G(l.wi));
    RC(linear_dgrad(c, dhmid, ff, M, ff, P[l.wi].warena_off, d, e->dxn, KLAB_F32));
    {
      const uint32_t tprev = dec ? tag_of(stack_id, i, SITE_XOUT) : tag_of(stack_id, i, SITE_ATTN_OUT);
      dy = next_dy();
      RC(rms_bwd(e->dxn, s.h[j], W[l.ln2], b.rstd3, dh_cur, dh_oth, dy, l.ln2, 0.f, 0, p, tprev));
      float* t = dh_cur; dh_cur = dh_oth; dh_oth = t;
    }
    if (dec) {  // ---------------- cross attention ----------------
      --j;
      void* dqc = e->dqc_pool[i];
      wq.push(dy, d, b.ctx2, inner, M, d, inner, G(l.co));
      RC(linear_dgrad(c, dy, d, M, d, P[l.co].warena_off, inner, e->dctx, c.dt));
      klab_attn_args a;
      memset(&a, 0, sizeof(a));
      a.dtype = c.dt; a.q = b.qc; a.ldq = inner;
      a.k = eoff(c, (void*)kv_all, (long)i * 2 * inner); a.ldk = kv_ld;
      a.v = eoff(c, (void*)kv_all, (long)i * 2 * inner + inner); a.ldv = kv_ld;
      a.ctx = b.ctx2; a.ldo = inner; a.lse = b.lse2; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lkv; a.dk = dk;
      a.drop_p = p; a.seed_dev = e->seed_dev; a.drop_tag = tag_of(stack_id, i, SITE_XPROB);
      a.dctx = e->dctx; a.lddo = inner; a.dq = dqc; a.lddq = inner;
      a.dk_out = eoff(c, dkv_all, (long)i * 2 * inner); a.lddk = kv_ld;
      a.dv = eoff(c, dkv_all, (long)i * 2 * inner + inner); a.lddv = kv_ld;
      RC(klab_t5_attn_bwd(&a, c.ws()));
      if (kv_in_queue) {  // this layer's k | v projection: its slice of d(k|v) is final, the product joins the layer's group
        wq.push(eoff(c, dkv_all, (long)i * 2 * inner), kv_ld, e->enc.out_t, d, B * Lkv, 2 * inner, d, Gflat + e->kvall_g_off + (long)i * 2 * inner * d);
        e->kv_wgrad_done = true;
      }
      wq.push(dqc, inner, b.xn2, d, M, inner, d, G(l.cq));
      RC(linear_dgrad(c, dqc, inner, M, inner, P[l.cq].warena_off, d, e->dxn, KLAB_F32));
      dy = next_dy();
      RC(rms_bwd(e->dxn, s.h[j], W[l.ln1], b.rstd2, dh_cur, dh_oth, dy, l.ln1, 0.f, 0, p, tag_of(stack_id, i, SITE_ATTN_OUT)));
      float* t = dh_cur; dh_cur = dh_oth; dh_oth = t;
    }
    // ---------------- self attention ----------------
    --j;
    wq.push(dy, d, b.ctx, inner, M, d, inner, G(l.o));
    RC(linear_dgrad(c, dy, d, M, d, P[l.o].warena_off, inner, e->dctx, c.dt));
    {
      klab_attn_args a;
      memset(&a, 0, sizeof(a));
      a.dtype = c.dt; a.q = b.qkv; a.ldq = 3 * inner; a.k = eoff(c, b.qkv, inner); a.ldk = 3 * inner;
      a.v = eoff(c, b.qkv, 2 * inner); a.ldv = 3 * inner; a.bias = s.bias; a.causal = dec ? 1 : 0;
      a.ctx = b.ctx; a.ldo = inner; a.lse = b.lse; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lq; a.dk = dk;
      a.drop_p = p; a.seed_dev = e->seed_dev; a.drop_tag = tag_of(stack_id, i, SITE_PROB);
      a.dctx = e->dctx; a.lddo = inner; a.dq = dqkv; a.lddq = 3 * inner;
      a.dk_out = eoff(c, dqkv, inner); a.lddk = 3 * inner; a.dv = eoff(c, dqkv, 2 * inner); a.lddv = 3 * inner;
      a.dbias = s.dbias;
      int arc = KLAB_ERR_UNSUPPORTED;
      if (c.dt == KLAB_BF16 && deferred == (int)L.size() - 1 - i) {  // store dS per layer; ONE reduction after the stack
        a.ds_defer = 1;
        a.ds_ws = (char*)e->ds_ws + (size_t)deferred * B * H * Lq * ((Lq + 31) & ~31) * c.es;
        arc = klab_t5_attn_bwd(&a, c.ws());
        if (arc == 0) ++deferred;
      }
      if (arc == KLAB_ERR_UNSUPPORTED) { a.ds_defer = 0; a.ds_ws = nullptr; arc = klab_t5_attn_bwd(&a, c.ws()); }
      RC(arc);
    }
    wq.push(dqkv, 3 * inner, b.xn1, d, M, 3 * inner, d, G(l.q));  // q|k|v grads are adjacent
    RC(linear_dgrad(c, dqkv, 3 * inner, M, 3 * inner, P[l.q].warena_off, d, e->dxn, KLAB_F32));
    {
      const bool first = (i == 0);
      if (!first) dy = next_dy();
      RC(rms_bwd(e->dxn, s.h[j], W[l.ln0], b.rstd1, dh_cur, dh_oth, first ? nullptr : dy, l.ln0, 0.f, 0, first ? 0.f : p,
                 first ? 0u : tag_of(stack_id, i - 1, SITE_FFN_OUT)));
      float* t = dh_cur; dh_cur = dh_oth; dh_oth = t;
    }
    bool launched = false;
    RC(wq.flush(c, flush_at[i] != 0, &launched));  // (one main->side event per release; the launch overlaps the following layers' chain)
    pending_buckets.push_back((int)L.size() - 1 - i);
    if (launched) {  // the buckets of every layer released so far are final behind this point of the side stream
      const int seg = dec ? 0 : 1;
      for (int bi : pending_buckets)
        if (e->bucket_events_on && Gflat == e->G[2] && !e->use_graph && bi < (int)e->bucket_ev[seg].size())
          RC((int)hipEventRecord(e->bucket_ev[seg][bi], e->side));
      pending_buckets.clear();
      if (i == 0 && e->bucket_events_on && Gflat == e->G[2] && !e->use_graph && (int)L.size() - 1 < (int)e->bucket_ev[seg].size())
        e->bucket_ev_live[seg] = true;
    }
  }
  if (part_ok && rms_calls) {
    if (rms_calls != e->rms_ncalls[stk]) return KLAB_ERR_BADARG;  // the destination table was built for exactly this visiting order
    RC(klab_colpart_reduce(e->rms_part, e->rms_part_stride, klab_rmsnorm_part_rows(M), d, e->rms_dst_dev[stk], rms_calls, c.ws()));
  }
  if (deferred) RC(klab_dbias_reduce(e->ds_ws, c.dt, s.dbias, deferred * B, H, Lq, Lq, c.ws()));
  RC(klab_relbias_bwd(s.dbias, s.bucket, G(L[0].relb), H, Lq, Lq, cfg.rel_buckets, c.ws()));
  *dh_out = dh_cur;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Swin-V2 forward (HF/swinv2:917-958, eval mode always: SURVEY §0.4)
// ------------------------------------------------------------------------------------------------
int swin_forward(const Ctx& c, const float* pixels, float p_in, bool refresh_bias) {
  klab_engine* e = c.e;
  const klab_swin_cfg& s = e->cfg.swin;
  const auto& P = e->P[0];
  const auto& W = e->W[0];
  const int B = e->B, R0 = s.image_size / s.patch, K0 = s.in_ch * s.patch * s.patch, C0 = s.embed_dim;
  const long M0 = (long)B * R0 * R0;
  // patch embedding (Conv2d k4 s4, HF/swinv2:281) = im2col + GEMM.  bf16: both operands' rows are zero-padded from K0 = 48
  // to 64 columns so that the GEMM takes the LDS-DMA path (K % 32 == 0); zeros meet zeros.
  const int Kp = e->pe_kp;
  (void)K0;
  // frozen tower: convolution + LayerNorm in one launch straight from the pixels (no column matrix, nothing kept for a backward)
  static const bool fused_pe = [] { const char* v = getenv("KLAB_SWIN_FUSED_EMBED"); return !v || atoi(v) != 0; }();
  int perc = KLAB_ERR_UNSUPPORTED;
  if (!e->cfg.train_swin && fused_pe && !e->fp8 && Kp >= 64)
    perc = klab_swin_patch_embed_fused(pixels, woff(c, P[e->si.pew].warena_off), Kp, W[e->si.peb], W[e->si.penw], W[e->si.penb], e->x0, e->x0t, c.dt,
                                       B, s.in_ch, s.image_size, s.patch, C0, s.ln_eps, c.ws());
  if (perc != 0 && perc != KLAB_ERR_UNSUPPORTED) return perc;
  if (perc != 0) {
  RC(klab_im2col_patch_ld(pixels, e->cols, c.dt, B, s.in_ch, s.image_size, s.patch, Kp, c.ws()));
  {
    klab_gemm_args g = G0(c, (int)M0, C0, Kp, e->cols, Kp, 1, woff(c, P[e->si.pew].warena_off), Kp, 1, e->pe_out, C0, c.dt);
    g.bias = W[e->si.peb];
    RC(klab_gemm(&g, c.ws()));
  }
  RC(klab_layernorm_fwd(e->pe_out, c.dt, W[e->si.penw], W[e->si.penb], nullptr, e->x0, e->x0t, c.dt, e->pe_mean, e->pe_rstd, (int)M0, C0,
                        s.ln_eps, 0, 0, 0, 0.f, nullptr, 0, c.ws()));
  }
  float* x = e->x0;
  void* xt = e->x0t;
  long bias_off = 0;
  for (int st = 0; st < s.n_stages; ++st) {
    SwinStageBufs& sb = e->sw[st];
    for (size_t k = 0; k < sb.blk.size(); ++k) {
      SwinBlockBufs& q = sb.blk[k];
      const SwinBlockIdx& ix = e->si.st[st].blk[k];
      const int C = q.C, M = (int)q.M, F = s.mlp_ratio * C, n = q.w * q.w;
      q.x_in = x; q.xt_in = xt;
      if (refresh_bias) {  // input-independent: with a frozen Swin the 16*sigmoid(CPB) tables are computed once per weight version
        if (q.btab)
          RC(klab_swin_cpb_table(e->swin_coords[st], W[ix.c0w], W[ix.c0b], W[ix.c2w], q.table, q.hidden, q.btab, e->swin_ntab[st], q.H, 512,
                                 c.ws()));
        else
          RC(klab_swin_cpb_bias(e->swin_coords[st], e->swin_index[st], W[ix.c0w], W[ix.c0b], W[ix.c2w], q.table, q.hidden, q.bias,
                                e->swin_ntab[st], n, q.H, 512, c.ws()));
      }
      const float* qkvb = ix.qb >= 0 ? e->farena + bias_off : nullptr;
      bias_off += 3 * C;
      static const bool fused_qkv = [] { const char* v = getenv("KLAB_SWIN_FUSED_QKV"); return !v || atoi(v) != 0; }();
      int qrc = KLAB_ERR_UNSUPPORTED;
      if (!e->cfg.train_swin && fused_qkv && q.bias)  // frozen tower, narrow stage, one-tile window: q|k|v never leave the chip
        qrc = klab_swin_qkv_attn_fused(xt, woff(c, P[ix.qw].warena_off), qkvb, q.ctx, q.bias, W[ix.ls], c.dt, B, q.R, q.w, q.shift, q.H, C,
                                       c.ws());
      if (qrc != 0 && qrc != KLAB_ERR_UNSUPPORTED) return qrc;
      if (qrc != 0) {
        RC(linear_fwd(c, xt, M, C, P[ix.qw].warena_off, 3 * C, q.qkv, 3 * C, c.dt, qkvb));
        klab_swin_attn_args a;
        memset(&a, 0, sizeof(a));
        a.dtype = c.dt; a.qkv = q.qkv; a.ctx = q.ctx; a.bias = q.bias; a.bias_table = q.btab; a.logit_scale = W[ix.ls]; a.lse = q.lse;
        a.B = B; a.R = q.R; a.w = q.w; a.shift = q.shift; a.H = q.H; a.C = C;
        a.bwd_ws = e->sattn_ws; a.bwd_ws_bytes = e->sattn_ws_bytes;  // (large windows: window-major copies for the streaming kernel)
        a.v_bias = qkvb ? qkvb + 2 * C : nullptr;                     // (padded windows: rows of the padded keys)
        RC(klab_swin_attn_fwd(&a, c.ws()));
      }
      static const bool fused_proj = [] { const char* v = getenv("KLAB_SWIN_FUSED_PROJ"); return !v || atoi(v) != 0; }();
      int prc = KLAB_ERR_UNSUPPORTED;
      if (!e->cfg.train_swin && fused_proj)  // frozen tower, narrow stage: output projection + LayerNorm + residual in one kernel
        prc = klab_swin_proj_ln_fused(q.ctx, x, woff(c, P[ix.pw].warena_off), W[ix.pb], W[ix.ln1w], W[ix.ln1b], q.h1, q.h1t, c.dt, M, C, s.ln_eps,
                                      c.ws());
      if (prc != 0 && prc != KLAB_ERR_UNSUPPORTED) return prc;
      // frozen tower, wide stage (C = 256): Linear + bias + LayerNorm + residual in one launch, 64 rows x all columns per workgroup
      static const bool fused_lin_ln = [] { const char* v = getenv("KLAB_SWIN_FUSED_LIN_LN"); return !v || atoi(v) != 0; }();
      const bool wide_ln = !e->cfg.train_swin && !e->fp8 && fused_lin_ln;
      if (prc != 0 && wide_ln) {
        prc = klab_swin_linear_ln_fused(q.ctx, x, woff(c, P[ix.pw].warena_off), W[ix.pb], W[ix.ln1w], W[ix.ln1b], q.h1, q.h1t, c.dt, M, C, C,
                                        s.ln_eps, c.ws());
        if (prc != 0 && prc != KLAB_ERR_UNSUPPORTED) return prc;
      }
      if (prc != 0) {
        RC(linear_fwd(c, q.ctx, M, C, P[ix.pw].warena_off, C, q.po, C, c.dt, W[ix.pb]));
        RC(ln_fwd_for_linear(c, q.po, W[ix.ln1w], W[ix.ln1b], x, q.h1, q.h1t, q.mean1, q.rstd1, M, C, s.ln_eps));
      }
      if (e->cfg.train_swin) {  // keep the pre-activation for gelu'
        RC(linear_fwd(c, q.h1t, M, C, P[ix.f1w].warena_off, F, q.z, F, c.dt, W[ix.f1b]));
        RC(gelu_fwd_for_linear(c, q.z, q.a, M, F));
      } else {
        // frozen tower, narrow stage: fc1 + GELU + fc2 + LayerNorm + residual in one kernel (the hidden layer stays on chip)
        static const bool fused_mlp = [] { const char* v = getenv("KLAB_SWIN_FUSED_MLP"); return !v || atoi(v) != 0; }();
        const int frc = !fused_mlp ? KLAB_ERR_UNSUPPORTED : klab_swin_mlp_fused(q.h1t, q.h1, woff(c, P[ix.f1w].warena_off), W[ix.f1b], woff(c, P[ix.f2w].warena_off), W[ix.f2b],
                                            W[ix.ln2w], W[ix.ln2b], q.h2, q.h2t, c.dt, M, C, s.ln_eps, c.ws());
        if (frc == 0) { x = q.h2; xt = q.h2t; continue; }
        if (frc != KLAB_ERR_UNSUPPORTED) return frc;
        RC(linear_fwd(c, q.h1t, M, C, P[ix.f1w].warena_off, F, q.a, F, c.dt, W[ix.f1b], KLAB_ACT_GELU));
      }
      int f2rc = KLAB_ERR_UNSUPPORTED;
      if (wide_ln)
        f2rc = klab_swin_linear_ln_fused(q.a, q.h1, woff(c, P[ix.f2w].warena_off), W[ix.f2b], W[ix.ln2w], W[ix.ln2b], q.h2, q.h2t, c.dt, M, F, C,
                                         s.ln_eps, c.ws());
      if (f2rc != 0 && f2rc != KLAB_ERR_UNSUPPORTED) return f2rc;
      if (f2rc != 0) {
        RC(linear_fwd(c, q.a, M, F, P[ix.f2w].warena_off, C, q.fo, C, c.dt, W[ix.f2b]));
        RC(ln_fwd_for_linear(c, q.fo, W[ix.ln2w], W[ix.ln2b], q.h1, q.h2, q.h2t, q.mean2, q.rstd2, M, C, s.ln_eps));
      }
      x = q.h2; xt = q.h2t;
    }
    if (st < s.n_stages - 1) {
      const int R = R0 >> st, C = C0 << st;
      const long M2 = (long)B * (R / 2) * (R / 2);
      RC(klab_merge_gather(x, sb.mg, c.dt, B, R, C, c.ws()));
      RC(linear_fwd(c, sb.mg, (int)M2, 4 * C, P[e->si.st[st].redw].warena_off, 2 * C, sb.mo, 2 * C, c.dt));
      RC(klab_layernorm_fwd(sb.mo, c.dt, W[e->si.st[st].mnw], W[e->si.st[st].mnb], nullptr, sb.xm, sb.xmt, c.dt, sb.mmean, sb.mrstd, (int)M2,
                            2 * C, s.ln_eps, 0, 0, 0, 0.f, nullptr, 0, c.ws()));
      x = sb.xm; xt = sb.xmt;
    }
  }
  // final LayerNorm (HF/swinv2:953) written straight into rows [0, N_img) of the T5 encoder input,
  // with the encoder's input dropout (HF/t5:725) applied on the way: ref/models/model.py:23 for free.
  const int Cl = C0 << (s.n_stages - 1);
  RC(klab_layernorm_fwd(x, KLAB_F32, W[e->si.lnw], W[e->si.lnb], nullptr, e->enc.h[0], nullptr, c.dt, e->sw_fmean, e->sw_frstd,
                        B * e->Nimg, Cl, s.ln_eps, e->Nimg, e->Le, 0, p_in, e->seed_dev, tag_of(STACK_ENC, 0, SITE_IN), c.ws()));
  return 0;
}

int swin_backward(const Ctx& c, const float* dh0, float p_in);

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" klab_engine* klab_engine_create(const klab_model_cfg* cfg) {
  if (!cfg) return nullptr;
  if (cfg->dtype != KLAB_F32 && cfg->dtype != KLAB_BF16 && cfg->dtype != KLAB_FP8) return nullptr;
  const klab_swin_cfg& s = cfg->swin;
  if (s.n_stages < 1 || s.n_stages > 8 || s.patch <= 0 || s.image_size % s.patch) return nullptr;
  const long Cl = (long)s.embed_dim << (s.n_stages - 1);
  // no projection between the towers: ref/models/model.py:23 concatenates on the sequence axis
  if (Cl != cfg->main.d_model || cfg->lang.d_model != cfg->main.d_model) return nullptr;
  klab_engine* e = new klab_engine();
  e->cfg = *cfg;
  if (cfg->dtype == KLAB_FP8) { e->fp8 = true; e->cfg.dtype = KLAB_BF16; }  // storage, backward and every non-GEMM kernel: bf16
  build_swin_params(cfg->swin, e->P[0], e->si);
  build_t5_params(cfg->lang, true, e->P[1], e->li);
  build_t5_params(cfg->main, false, e->P[2], e->mi);
  plan_arenas(e);
  plan_grads(e);
  return e;
}

// Work the engine enqueued may still be running: on its own streams, and -- launched graphs, eager kernels -- on whatever
// stream the caller passed last.  Nothing the engine owns is released before the device has drained it.
static void drain_engine(klab_engine* e) {
  if (!e->side && !e->own) return;  // never bound: nothing was ever enqueued
  if (e->own) hipStreamSynchronize(e->own);
  if (e->side) hipStreamSynchronize(e->side);
  hipDeviceSynchronize();  // the caller's stream(s); destroy / rebind are rare, a device-wide wait is the simple safe form
}

extern "C" void klab_engine_destroy(klab_engine* e) {
  if (!e) return;
  drain_engine(e);
  for (auto& g : e->gs) if (g.exec) { hipGraphExecDestroy(g.exec); g.exec = nullptr; }
  if (e->ev_fork) hipEventDestroy(e->ev_fork);
  if (e->ev_join) hipEventDestroy(e->ev_join);
  if (e->ev_zero) hipEventDestroy(e->ev_zero);
  if (e->ev_in) hipEventDestroy(e->ev_in);
  if (e->ev_out) hipEventDestroy(e->ev_out);
  for (auto ev : e->evpool) if (ev) hipEventDestroy(ev);
  for (auto& v : e->bucket_ev) for (auto ev : v) if (ev) hipEventDestroy(ev);
  for (auto ev : e->swin_done_ev) if (ev) hipEventDestroy(ev);
  for (auto& pr : e->probe) {
    for (auto ev : pr.a) if (ev) hipEventDestroy(ev);
    for (auto ev : pr.b) if (ev) hipEventDestroy(ev);
  }
  if (e->side) hipStreamDestroy(e->side);  // streams last: every event recorded on them is gone
  if (e->own) hipStreamDestroy(e->own);
  delete e;
}

extern "C" int klab_engine_num_params(const klab_engine* e, int model) {
  if (!e || model < 0 || model > 2) return -1;
  return (int)e->P[model].size();
}

extern "C" int klab_engine_param_info(const klab_engine* e, int model, int i, char* name, int name_cap, long* shape4, int* ndim,
                                      long* grad_off) {
  if (!e || model < 0 || model > 2 || i < 0 || i >= (int)e->P[model].size()) return KLAB_ERR_BADARG;
  const ParamInfo& p = e->P[model][i];
  if (name && name_cap > 0) { strncpy(name, p.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (ndim) *ndim = (int)p.shape.size();
  if (shape4) for (size_t k = 0; k < 4; ++k) shape4[k] = k < p.shape.size() ? p.shape[k] : 1;
  if (grad_off) *grad_off = p.grad_off;
  return 0;
}

extern "C" long klab_engine_grad_elems(const klab_engine* e, int model) { return (e && model >= 0 && model <= 2) ? e->grad_elems[model] : -1; }

extern "C" int klab_engine_segment(const klab_engine* e, int seg, int* model, long* off, long* len) {
  if (!e || seg < 0 || seg > 2) return KLAB_ERR_BADARG;
  if (model) *model = seg == 2 ? 0 : 2;
  if (off) *off = e->seg_off[seg];
  if (len) *len = e->seg_len[seg];
  return 0;
}

extern "C" int klab_engine_num_buckets(const klab_engine* e, int segment) {
  if (!e || segment < 0 || segment > 2) return -1;
  return (int)e->buckets[segment].size();
}
extern "C" int klab_engine_bucket(const klab_engine* e, int segment, int i, long* off, long* len) {
  if (!e || segment < 0 || segment > 2 || i < 0 || i >= (int)e->buckets[segment].size()) return KLAB_ERR_BADARG;
  if (off) *off = e->buckets[segment][i].off;
  if (len) *len = e->buckets[segment][i].len;
  return KLAB_OK;
}
extern "C" int klab_engine_set_bucket_events(klab_engine* e, int on) {
  if (!e) return KLAB_ERR_BADARG;
  e->bucket_events_on = on != 0;
  if (!on) for (auto& b : e->bucket_ev_live) b = false;
  return KLAB_OK;
}
extern "C" int klab_engine_bucket_wait(klab_engine* e, int segment, int i, void* stream) {
  if (!e || !e->bound || segment < 0 || segment > 2 || i < 0 || i >= (int)e->bucket_ev[segment].size()) return KLAB_ERR_BADARG;
  if (!e->bucket_ev_live[segment]) return KLAB_ERR_UNSUPPORTED;  // graph replay, or no backward of that segment yet
  return (int)hipStreamWaitEvent((hipStream_t)stream, e->bucket_ev[segment][i], 0);
}

extern "C" size_t klab_engine_workspace_bytes(klab_engine* e, int B, int Ls, int Lt) {
  if (!e || B <= 0 || Ls <= 0 || Lt <= 0) return 0;
  // plan on a scratch engine that carries only what the plan reads (configuration, table sizes, arena extents): the bound
  // state of `e` -- and its live stream / event / graph handles -- is neither touched nor copied
  klab_engine tmp;
  tmp.cfg = e->cfg;
  for (int m = 0; m < 3; ++m) tmp.P[m].resize(e->P[m].size());
  tmp.warena_elems = e->warena_elems; tmp.farena_elems = e->farena_elems; tmp.fp8 = e->fp8;
  return plan_workspace(&tmp, nullptr, B, Ls, Lt);
}

extern "C" int klab_engine_bind(klab_engine* e, int B, int Ls, int Lt, void* workspace, size_t ws_bytes, const void* const* swin_params,
                                const void* const* lang_params, const void* const* main_params, float* main_grads, float* swin_grads,
                                const int* lang_bucket, const int* enc_bucket, const int* dec_bucket, const void* const* swin_coords,
                                const void* const* swin_index, void* stream) {
  if (!e || !workspace || !swin_params || !lang_params || !main_params || !main_grads) return KLAB_ERR_BADARG;
  if (e->cfg.train_swin && !swin_grads) return KLAB_ERR_BADARG;
  // a rebind (new batch shape / device): kernels and captured graphs of the previous binding may still be in flight and
  // address the old workspace, which the caller releases after this call
  // The dropout RNG of the model (base seed, forwards since seeding) lives in device words of the binding's workspace; a rebind
  // must not restart it -- the reference loop pads every batch to its longest row (ref/train.py:56-57), i.e. rebinds almost every
  // step, and a restarted counter would replay the masks of step 1 each time.  Carried over to the new workspace below.
  uint32_t carry[3] = {0, 0, 0};
  bool carry_rng = false;
  if (e->bound) {
    e->bound = false;
    drain_engine(e);
    if (e->seed_set && e->seed_dev) carry_rng = hipMemcpy(carry, e->seed_dev, sizeof(carry), hipMemcpyDeviceToHost) == hipSuccess;
  }
  const size_t need = plan_workspace(e, workspace, B, Ls, Lt);
  if (need > ws_bytes) return KLAB_ERR_BADARG;
  const void* const* src[3] = {swin_params, lang_params, main_params};
  for (int m = 0; m < 3; ++m) {
    e->W[m].resize(e->P[m].size());
    for (size_t i = 0; i < e->P[m].size(); ++i) {
      e->W[m][i] = (const float*)src[m][i];
      if (!e->W[m][i]) return KLAB_ERR_BADARG;
    }
  }
  e->G[0] = swin_grads; e->G[2] = main_grads;
  e->lang_bucket = lang_bucket; e->enc_bucket = enc_bucket; e->dec_bucket = dec_bucket;
  e->lang.bucket = lang_bucket; e->enc.bucket = enc_bucket; e->dec.bucket = dec_bucket;
  const klab_swin_cfg& s = e->cfg.swin;
  e->swin_coords.resize(s.n_stages); e->swin_index.resize(s.n_stages); e->swin_ntab.resize(s.n_stages);
  const int R0 = s.image_size / s.patch;
  for (int st = 0; st < s.n_stages; ++st) {
    e->swin_coords[st] = (const float*)swin_coords[st];
    e->swin_index[st] = (const int*)swin_index[st];
    const int R = R0 >> st, w = R < s.window ? R : s.window;
    if (w <= 0) return KLAB_ERR_UNSUPPORTED;
    e->swin_ntab[st] = (2 * w - 1) * (2 * w - 1);
  }
  // cast descriptors: {src, dst_off, n4_prefix}
  hipStream_t hs = (hipStream_t)stream;
  for (int grp = 0; grp < 2; ++grp) {  // 0: trainable (main, and Swin when --image_model_train), 1: frozen (lang, frozen Swin)
    std::vector<long> d;
    long pre = 0; int n = 0;
    for (int m = 0; m < 3; ++m) {
      const bool frozen = (m == 1) || (m == 0 && !e->cfg.train_swin);
      if (frozen != (grp == 1)) continue;
      for (size_t i = 0; i < e->P[m].size(); ++i) {
        const ParamInfo& p = e->P[m][i];
        if (p.warena_off < 0) continue;
        if (p.numel % 4) return KLAB_ERR_UNSUPPORTED;
        if (m == 0 && (int)i == e->si.pew && e->pe_kp != e->pe_k0) {  // row by row into the padded pitch
          if (e->pe_k0 % 4) return KLAB_ERR_UNSUPPORTED;
          for (int r = 0; r < e->cfg.swin.embed_dim; ++r) {
            d.push_back((long)(e->W[m][i] + (long)r * e->pe_k0)); d.push_back(p.warena_off + (long)r * e->pe_kp); d.push_back(pre);
            pre += e->pe_k0 / 4; ++n;
          }
          continue;
        }
        d.push_back((long)e->W[m][i]); d.push_back(p.warena_off); d.push_back(pre);
        pre += p.numel / 4; ++n;
      }
    }
    const size_t grp_cap = e->P[0].size() + e->P[1].size() + e->P[2].size() + e->cfg.swin.embed_dim + 1;
    void* dst = grp == 0 ? e->cast_desc : (void*)((char*)e->cast_desc + sizeof(long) * 3 * grp_cap);
    if (grp == 1) { e->cast_desc_frozen = dst; e->n_cast_frozen = n; e->cast_total4_frozen = pre; }
    if (n == 0) { if (grp == 0) { e->n_cast = 0; e->cast_total4 = 0; } continue; }
    hipError_t er0 = hipMemcpyAsync(dst, d.data(), d.size() * sizeof(long), hipMemcpyHostToDevice, hs);
    if (er0 != hipSuccess) return (int)er0;
    er0 = hipStreamSynchronize(hs);  // bind time only (d goes out of scope)
    if (er0 != hipSuccess) return (int)er0;
    if (grp == 1) continue;
    e->n_cast = n; e->cast_total4 = pre;
  }
  e->frozen_valid = false;
  e->seg1_zeroed = e->dbias_zeroed[0] = e->dbias_zeroed[1] = e->denc_in_dxn = false;  // (hand-offs between backward segments of the OLD binding)
  e->bias_ready[0] = e->bias_ready[1] = e->bias_ready[2] = e->dec_embed_ready = e->inv_n_ready = e->kv_wgrad_done = false;
  e->loss_out = nullptr;
  if (e->pe_kp != e->pe_k0)  // the padding columns of the patch-embedding weight rows: written here, never again
    RC((int)hipMemsetAsync((char*)e->warena + (size_t)e->P[0][e->si.pew].warena_off * e->es, 0,
                           (size_t)e->cfg.swin.embed_dim * e->pe_kp * e->es, hs));
  {
    std::vector<long> d;
    long pre = 0; int n = 0;
    for (size_t i = 0; i < e->P[0].size(); ++i) {
      const ParamInfo& p = e->P[0][i];
      if (p.farena_off < 0) continue;
      if (p.numel % 4) return KLAB_ERR_UNSUPPORTED;
      d.push_back((long)e->W[0][i]); d.push_back(p.farena_off); d.push_back(pre);
      pre += p.numel / 4; ++n;
    }
    e->n_fcast = n; e->fcast_total4 = pre;
    RC((int)hipMemsetAsync(e->farena, 0, (size_t)e->farena_elems * 4, hs));
    if (n) {
      hipError_t er = hipMemcpyAsync(e->fcast_desc, d.data(), d.size() * sizeof(long), hipMemcpyHostToDevice, hs);
      if (er != hipSuccess) return (int)er;
      er = hipStreamSynchronize(hs);
      if (er != hipSuccess) return (int)er;
    }
  }
  if (e->fp8) {  // fp8 weight table: every arena tensor as rows x K (the padded patch-embedding weight stays bf16)
    std::vector<long> d;
    long row0 = 0; int n = 0;
    for (int m = 0; m < 3; ++m)
      for (size_t i = 0; i < e->P[m].size(); ++i) {
        const ParamInfo& p = e->P[m][i];
        if (p.warena_off < 0 || p.shape.empty() || (m == 0 && (int)i == e->si.pew)) continue;
        const long rows = p.shape[0], K = p.numel / rows;
        if (!fp8_weight_ok(K) || (p.warena_off & 15)) continue;
        d.push_back(p.warena_off); d.push_back(rows); d.push_back(K); d.push_back(row0);
        row0 += rows; ++n;
      }
    e->n_qdesc = n; e->q_rows = row0;
    if (n) {
      hipError_t er = hipMemcpyAsync(e->qdesc, d.data(), d.size() * sizeof(long), hipMemcpyHostToDevice, hs);
      if (er != hipSuccess) return (int)er;
      er = hipStreamSynchronize(hs);
      if (er != hipSuccess) return (int)er;
    }
  }
  {  // fused Adam descriptors: every trainable tensor of the main T5 (tied tables appear once)
    std::vector<long> d;
    long pre = 0; int n = 0;
    bool ok = true;
    // tensors of backward segment 0 first, then segment 1: a prefix range of the table = one segment (the optimizer may
    // update segment 0 while segment 1's gradient all-reduce is still in flight, klab_engine_adam_step_segment)
    for (int seg = 0; seg < 2 && ok; ++seg) {
      if (seg == 1) e->adam_split4 = pre;
      for (size_t i = 0; i < e->P[2].size(); ++i) {
        const ParamInfo& p = e->P[2][i];
        if (p.grad_off < 0 || (p.grad_off >= e->seg_off[1]) != (seg == 1)) continue;
        if (p.numel % 4 || (p.grad_off & 3) || (p.warena_off >= 0 && (p.warena_off & 3))) { ok = false; break; }
        d.push_back((long)e->W[2][i]); d.push_back(p.grad_off); d.push_back(p.warena_off); d.push_back(pre);
        pre += p.numel / 4; ++n;
      }
    }
    e->n_adam = ok ? n : 0; e->adam_total4 = pre;
    if (e->n_adam) {
      hipError_t er = hipMemcpyAsync(e->adam_desc, d.data(), d.size() * sizeof(long), hipMemcpyHostToDevice, hs);
      if (er != hipSuccess) return (int)er;
      er = hipStreamSynchronize(hs);
      if (er != hipSuccess) return (int)er;
    }
  }
  for (int st = 0; st < 2; ++st) {  // norm-weight gradient destinations in the order the stack's backward visits them
    const bool dec = st == 1;
    const auto& L = dec ? e->mi.dec : e->mi.enc;
    std::vector<float*> dst;
    float* Gm = e->G[2];
    dst.push_back(Gm + e->P[2][dec ? e->mi.dec_final : e->mi.enc_final].grad_off);
    for (int i = (int)L.size() - 1; i >= 0; --i) {
      dst.push_back(Gm + e->P[2][L[i].ln2].grad_off);
      if (dec) dst.push_back(Gm + e->P[2][L[i].ln1].grad_off);
      dst.push_back(Gm + e->P[2][L[i].ln0].grad_off);
    }
    e->rms_ncalls[st] = (int)dst.size();
    hipError_t er = hipMemcpyAsync(e->rms_dst_dev[st], dst.data(), dst.size() * sizeof(float*), hipMemcpyHostToDevice, hs);
    if (er != hipSuccess) return (int)er;
    er = hipStreamSynchronize(hs);
    if (er != hipSuccess) return (int)er;
  }
  RC((int)hipMemsetAsync(e->seed_dev, 0, 512, hs));
  for (auto& g : e->gs) {  // rebinding drops the captured graphs (drained above)
    if (g.exec) hipGraphExecDestroy(g.exec);
    g = klab_engine::GraphSlot();
  }
  e->seed_set = false;
  if (carry_rng) {  // (stream-ordered behind the memset above; seed_base is unchanged, so the next forward keeps the counter)
    RC((int)hipMemcpyAsync(e->seed_dev, carry, sizeof(carry), hipMemcpyHostToDevice, hs));
    RC((int)hipStreamSynchronize(hs));  // `carry` is a stack buffer
    e->seed_set = true;
  }
  if (!e->side) {
    {  // experiment knob: KLAB_SIDE_PRIO=low|high gives the side stream (weight gradients, language encoder) another priority
      const char* pv = getenv("KLAB_SIDE_PRIO");
      int lo = 0, hi = 0;
      // experiment knob: KLAB_SIDE_CUS=n restricts the side stream to n of the 256 CUs (n / 8 per XCD, CU-mask bits are dealt
      // round-robin over the XCDs) -- does confining the weight gradients to part of the chip hurt the main chain less?
      const char* cm = getenv("KLAB_SIDE_CUS");
      const int ncu = cm ? atoi(cm) : 0;
      if (ncu >= 8 && ncu < 256) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < ncu; ++i) mask[i >> 5] |= 1u << (i & 31);
        RC((int)hipExtStreamCreateWithCUMask(&e->side, 8, mask));
      } else if (pv && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && (pv[0] == 'l' || pv[0] == 'h'))
        RC((int)hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, pv[0] == 'l' ? lo : hi));
      else
        RC((int)hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
    }
    RC((int)hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    RC((int)hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    RC((int)hipEventCreateWithFlags(&e->ev_zero, hipEventDisableTiming));
    RC((int)hipStreamCreateWithFlags(&e->own, hipStreamNonBlocking));
    e->evpool.resize(256);
    for (auto& ev : e->evpool) RC((int)hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    RC((int)hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
    RC((int)hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming));
    for (auto& ev : e->swin_done_ev) RC((int)hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int sg = 0; sg < 3; ++sg) {
      e->bucket_ev[sg].assign(e->buckets[sg].size(), nullptr);
      for (auto& ev : e->bucket_ev[sg]) RC((int)hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
  }
  for (int sg = 0; sg < 3; ++sg) e->bucket_ev_live[sg] = false;
  e->bound = true;
  return 0;
}

namespace {

__global__ void seed_set_kernel(uint32_t* st, uint32_t base) { st[1] = base; st[2] = 0; }
__global__ void seed_restore_kernel(uint32_t* st, uint32_t base, uint32_t counter) { st[1] = base; st[2] = counter; }
__global__ void seed_step_kernel(uint32_t* st) {
  const uint32_t n = st[2] + 1;
  st[2] = n;
  uint32_t x = st[1] + 0x9E3779B1u * n;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  st[0] = x;
}

// the stream the engine really works on for this call, and the fences that tie it to the caller's stream
inline hipStream_t enter_stream(klab_engine* e, hipStream_t caller, int& rc) {
  rc = 0;
  if (!e->use_graph || caller != nullptr) return caller;
  rc = (int)hipEventRecord(e->ev_in, caller);
  if (!rc) rc = (int)hipStreamWaitEvent(e->own, e->ev_in, 0);
  return e->own;
}
inline int leave_stream(klab_engine* e, hipStream_t caller, hipStream_t used) {
  if (used == caller) return 0;
  int rc = (int)hipEventRecord(e->ev_out, used);
  if (!rc) rc = (int)hipStreamWaitEvent(caller, e->ev_out, 0);
  return rc;
}

template <typename F>
int run_graphed(klab_engine* e, int slot, hipStream_t s, F body) {
  klab_engine::GraphSlot& g = e->gs[slot];
  if (!e->use_graph || g.failed) return body();
  if (g.exec) return (int)hipGraphLaunch(g.exec, s);
  if (g.uses++ == 0) return body();  // first use eager: kernel attributes (dynamic LDS sizes) get set outside capture
  hipError_t er = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  if (er != hipSuccess) { g.failed = true; return body(); }
  const int rc = body();
  hipGraph_t graph = nullptr;
  er = hipStreamEndCapture(s, &graph);
  if (rc != 0 || er != hipSuccess || !graph) {
    if (graph) hipGraphDestroy(graph);
    g.failed = true;
    return rc != 0 ? rc : body();  // capture enqueued nothing: run the sequence for real
  }
  er = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (er != hipSuccess) { g.exec = nullptr; g.failed = true; return body(); }
  return (int)hipGraphLaunch(g.exec, s);
}

// everything of the forward before the LM head (inputs are the engine-owned staged copies)
int forward_part_a(klab_engine* e, hipStream_t stream, float p, bool refresh_frozen, bool trainable_current, bool encoder_current = false) {
  Ctx c{e, stream, e->cfg.dtype, e->es};
  const klab_model_cfg& cfg = e->cfg;
  const float* pixels = e->pixels_cur;
  const long long *src_ids = e->src_buf, *tgt_ids = e->tgt_buf;
  hipLaunchKernelGGL(seed_step_kernel, dim3(1), dim3(1), 0, c.s, e->seed_dev);
  // 1. weights: fp32 masters -> compute-dtype arena (+ fused f32 bias vectors)
  if (e->n_cast && !trainable_current) RC(klab_cast_pack(e->cast_desc, e->n_cast, e->cast_total4, e->warena, c.dt, c.ws()));
  if (refresh_frozen || e->cfg.train_swin) {
    if (refresh_frozen && e->n_cast_frozen)
      RC(klab_cast_pack(e->cast_desc_frozen, e->n_cast_frozen, e->cast_total4_frozen, e->warena, c.dt, c.ws()));
    if (e->n_fcast) RC(klab_cast_pack(e->fcast_desc, e->n_fcast, e->fcast_total4, e->farena, KLAB_F32, c.ws()));
  }
  // fp8 mode: e4m3 copies + per-row scales of every GEMM weight, from the bf16 arena (which the casts above or the fused
  // optimizer step just refreshed)
  if (e->fp8 && e->n_qdesc) RC(klab_quant_fp8_arena(e->qdesc, e->n_qdesc, e->q_rows, e->warena, e->w8, e->wscale, c.ws()));
  const int B = e->B, d = cfg.main.d_model;
  // 2./3. The two towers are independent until the concat.  Swin-V2 (ref/models/model.py:22, rows [0, N_img)) is
  //    enqueued FIRST on the main stream -- its launches are long, so the host runs far ahead -- and the frozen
  //    language encoder (model.py:20-21, rows [N_img, Le); ~100 launches of ~5 us over 576 tokens, which would
  //    otherwise be paced by the host) is then enqueued on the side stream and runs underneath it.
  if (!encoder_current) {  // (greedy decoding re-enters with the same images / prompt: encoder output and cross K/V stay valid)
  RC((int)hipEventRecord(e->ev_fork, c.s));
  RC(swin_forward(c, pixels, p, refresh_frozen || e->cfg.train_swin));
  RC((int)hipStreamWaitEvent(e->side, e->ev_fork, 0));
  {
    Ctx cs{e, e->side, e->cfg.dtype, e->es};
    // Small kernels that depend on nothing but the inputs and the weights run here, ahead of the language encoder and off the
    // main chain (ev_join covers them): the position biases of the encoder and decoder stacks and the decoder's input embedding
    // (5 launches of 4-6 us).  KLAB_EARLY_SMALL=0: where they were.
    static const bool early = [] { const char* v = getenv("KLAB_EARLY_SMALL"); return !v || atoi(v) != 0; }();
    if (early && !e->use_graph) {
      const int H = cfg.main.n_heads;
      RC(klab_relbias_fwd(e->W[2][e->mi.enc[0].relb], e->enc.bucket, e->enc.bias, H, e->enc.Lseq, e->enc.Lseq, cs.ws()));
      RC(klab_relbias_fwd(e->W[2][e->mi.dec[0].relb], e->dec.bucket, e->dec.bias, H, e->dec.Lseq, e->dec.Lseq, cs.ws()));
      e->bias_ready[STACK_ENC] = e->bias_ready[STACK_DEC] = true;
      RC(klab_embed_fwd(tgt_ids, 1, e->Lt, cfg.main.start_id, cfg.main.pad_id, e->W[2][e->mi.shared], cfg.main.vocab, e->dec.h[0], B * e->Lt, d, p,
                        e->seed_dev, tag_of(STACK_DEC, 0, SITE_IN), e->err_dev, cs.ws()));
      e->dec_embed_ready = true;
      RC(klab_ce_count(tgt_ids, B * e->Lt, e->inv_n, cs.ws()));  // 1 / n_valid of the labels: klab_ce_fwd then skips its counting launch
      e->inv_n_ready = true;
    }
    RC(klab_embed_fwd(src_ids, 0, e->Ls, 0, 0, e->W[1][e->li.shared], cfg.lang.vocab, e->lang.h[0], B * e->Ls, d, 0.f, nullptr, 0, e->err_dev,
                      cs.ws()));
    RC(t5_stack_forward(cs, cfg.lang, e->P[1], e->W[1], e->li.enc, e->li.enc_final, e->lang, false, STACK_LANG, 0.f, B, nullptr, 0, 0,
                        e->enc.h[0], e->Ls, e->Le, e->Nimg, p));
    RC((int)hipEventRecord(e->ev_join, e->side));
  }
  RC((int)hipStreamWaitEvent(c.s, e->ev_join, 0));
  // 4. T5 encoder (HF/t5:1009-1016)
  RC(t5_stack_forward(c, cfg.main, e->P[2], e->W[2], e->mi.enc, e->mi.enc_final, e->enc, false, STACK_ENC, p, B, nullptr, 0, 0, nullptr, 0, 0, 0,
                      p));
  }
  // 5. decoder: shift_right + embedding (HF/t5:1026-1028), cross K/V of all layers in one GEMM, stack
  const int inner = cfg.main.n_heads * cfg.main.d_kv, nld = cfg.main.n_dec_layers;
  if (e->dec_embed_ready) e->dec_embed_ready = false;
  else RC(klab_embed_fwd(tgt_ids, 1, e->Lt, cfg.main.start_id, cfg.main.pad_id, e->W[2][e->mi.shared], cfg.main.vocab, e->dec.h[0], B * e->Lt, d, p,
                         e->seed_dev, tag_of(STACK_DEC, 0, SITE_IN), e->err_dev, c.ws()));
  if (!encoder_current)
    RC(linear_fwd(c, e->enc.out_t, B * e->Le, d, e->kvall_w_off, nld * 2 * inner, e->kv_all, (long)nld * 2 * inner, c.dt));
  RC(t5_stack_forward(c, cfg.main, e->P[2], e->W[2], e->mi.dec, e->mi.dec_final, e->dec, true, STACK_DEC, p, B, e->kv_all, e->Le,
                      (long)nld * 2 * inner, nullptr, 0, 0, 0, p));
  return 0;
}

}  // namespace

extern "C" int klab_engine_adam_step(klab_engine* e, float* m, float* v, float lr, float beta1, float beta2, float eps, float weight_decay,
                                     float bias_corr1, float bias_corr2, void* stream) {
  if (!e || !e->bound || !m || !v || !e->G[2]) return KLAB_ERR_BADARG;
  if (!e->n_adam) return KLAB_ERR_UNSUPPORTED;
  return klab_adam_step(e->adam_desc, e->n_adam, e->adam_total4, e->G[2], m, v, e->warena, e->cfg.dtype, lr, beta1, beta2, eps, weight_decay,
                        bias_corr1, bias_corr2, stream);
}

extern "C" int klab_engine_adam_step_segment(klab_engine* e, int segment, float* m, float* v, float lr, float beta1, float beta2, float eps,
                                             float weight_decay, float bias_corr1, float bias_corr2, void* stream) {
  if (!e || !e->bound || !m || !v || !e->G[2] || segment < 0 || segment > 1) return KLAB_ERR_BADARG;
  if (!e->n_adam) return KLAB_ERR_UNSUPPORTED;
  const long b4 = segment == 0 ? 0 : e->adam_split4, e4 = segment == 0 ? e->adam_split4 : e->adam_total4;
  return klab_adam_step_range(e->adam_desc, e->n_adam, b4, e4, e->G[2], m, v, e->warena, e->cfg.dtype, lr, beta1, beta2, eps, weight_decay,
                              bias_corr1, bias_corr2, stream);
}

// Device-side dropout RNG of the binding: st[1] = base seed, st[2] = number of forwards since it was (re)seeded; every forward
// derives its masks from hash(base, counter).  A resumed run restores both so that it continues the mask stream instead of
// replaying it from step 1 (checkpoint.py).
extern "C" int klab_engine_get_rng(klab_engine* e, uint32_t* base, uint32_t* counter, void* stream) {
  if (!e || !e->bound || !base || !counter) return KLAB_ERR_BADARG;
  uint32_t st[3] = {0, 0, 0};
  RC((int)hipMemcpyAsync(st, e->seed_dev, sizeof(st), hipMemcpyDeviceToHost, (hipStream_t)stream));
  RC((int)hipStreamSynchronize((hipStream_t)stream));
  *base = e->seed_set ? st[1] : e->seed_base;
  *counter = e->seed_set ? st[2] : 0;
  return KLAB_OK;
}
extern "C" int klab_engine_set_rng(klab_engine* e, uint32_t base, uint32_t counter, void* stream) {
  if (!e || !e->bound) return KLAB_ERR_BADARG;
  hipLaunchKernelGGL(seed_restore_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, e->seed_dev, base, counter);
  e->seed_base = base; e->seed_set = true;  // the next forward with this base keeps the counter
  return KLAB_OK;
}

extern "C" int klab_engine_set_graph(klab_engine* e, int on) {
  if (!e) return KLAB_ERR_BADARG;
  e->use_graph = on != 0;
  return 0;
}

extern "C" int klab_engine_forward(klab_engine* e, const float* pixels, const long long* src_ids, const long long* tgt_ids, int training,
                                   uint32_t seed, int want_grad, void* stream) {
  if (!e || !e->bound || !pixels || !src_ids || !tgt_ids) return KLAB_ERR_BADARG;
  int erc = 0;
  const hipStream_t xs = enter_stream(e, (hipStream_t)stream, erc);
  if (erc) return erc;
  Ctx c{e, xs, e->cfg.dtype, e->es};
  const klab_model_cfg& cfg = e->cfg;
  // `training` bit 0: T5 dropout on; bit 1: the caller vouches that the frozen towers' weights are unchanged since the
  // previous forward of this binding (skip their re-cast and the Swin CPB tables)
  const float p = (training & 1) ? cfg.main.dropout : 0.f;
  const bool refresh_frozen = !((training & 2) && e->frozen_valid);
  e->p_train = p;
  e->last_tgt = e->tgt_buf;
  // stage the inputs (device-to-device, stream-ordered): replayed graphs read fixed addresses
  const int B = e->B, d = cfg.main.d_model;
  if (e->use_graph) {  // replayed graphs read fixed addresses; eager launches read the caller's tensor in place (only im2col does)
    RC((int)hipMemcpyAsync(e->pixels_buf, pixels, (size_t)B * cfg.swin.in_ch * cfg.swin.image_size * cfg.swin.image_size * 4,
                           hipMemcpyDeviceToDevice, c.s));
    e->pixels_cur = e->pixels_buf;
  } else {
    e->pixels_cur = pixels;
  }
  RC((int)hipMemcpyAsync(e->src_buf, src_ids, (size_t)B * e->Ls * 8, hipMemcpyDeviceToDevice, c.s));
  RC((int)hipMemcpyAsync(e->tgt_buf, tgt_ids, (size_t)B * e->Lt * 8, hipMemcpyDeviceToDevice, c.s));
  if (!e->seed_set || seed != e->seed_base) {  // (re)seed the device-side counter RNG; each forward then advances it itself
    hipLaunchKernelGGL(seed_set_kernel, dim3(1), dim3(1), 0, c.s, e->seed_dev, seed);
    e->seed_base = seed; e->seed_set = true;
  }
  // bit 2: the optimizer step (klab_engine_adam_step) already refreshed the trainable weights' compute-dtype copies.
  // Ignored under graph replay, whose captured sequence always contains the cast.
  const bool tcur = (training & 4) && !e->use_graph;
  // bit 3: same images, prompt and weights as the previous forward of this binding, evaluation mode: only the decoder and the
  // LM head run (greedy decoding, ref/models/model.py:28).  Ignored under graph replay and whenever gradients are wanted.
  const bool ecur = (training & 8) && !(training & 1) && !want_grad && !e->use_graph && e->frozen_valid && !refresh_frozen;
  if (refresh_frozen) {  // not worth a graph slot: happens once per weight version
    RC(forward_part_a(e, c.s, p, true, tcur));
    e->frozen_valid = true;
  } else if (ecur) {
    RC(forward_part_a(e, c.s, p, false, true, true));
  } else {
    RC(run_graphed(e, (training & 1) ? 1 : 0, c.s, [&]() { return forward_part_a(e, c.s, p, false, tcur); }));
  }
  {
    const int Md = B * e->Lt, V = cfg.main.vocab;
    klab_gemm_args g = G0(c, Md, V, d, e->dec.out_t, d, 1, woff(c, e->P[2][e->mi.shared].warena_off), d, 1, e->logits, V, c.dt);
    g.alpha = cfg.main.scale_decoder_outputs ? 1.f / sqrtf((float)d) : 1.f;  // HF/t5:1044-1045
    g.name_tag = 1;
    // the LM-head launch stays outside the graphs so that the probe's HIP events can bracket it
    klab_engine::Probe& pr = e->probe[0];
    const bool probe = e->probe_on && pr.n < (int)pr.a.size();
    if (probe) { klab::tl_launch_probe.a = pr.a[pr.n]; klab::tl_launch_probe.b = pr.b[pr.n]; }
    RC(fwd_gemm(c, g, e->P[2][e->mi.shared].warena_off));
    if (probe) {
      if (klab::tl_launch_probe.a) klab::tl_launch_probe.a = nullptr;  // (a launch path without the hook, e.g. fp8: not recorded)
      else pr.flops[pr.n++] = 2.0 * Md * (double)V * d;
    }
    const bool counted = e->inv_n_ready && !e->use_graph;
    e->inv_n_ready = false;
    float* loss_dst = e->loss_out && !e->use_graph ? e->loss_out : e->loss;  // (replayed graphs write the fixed address)
    e->loss_out = nullptr;
    RC(run_graphed(e, want_grad ? 3 : 2, c.s, [&]() {
      return klab_ce_fwd(e->logits, V, c.dt, e->tgt_buf, Md, V, e->inv_n, e->loss_row, loss_dst, (want_grad ? 1 : 0) | (counted ? 2 : 0), c.ws());
    }));
  }
  return leave_stream(e, (hipStream_t)stream, xs);
}

extern "C" int klab_engine_probe_enable(klab_engine* e, int on) {
  if (!e) return KLAB_ERR_BADARG;
  if (on && e->probe[0].a.empty()) {
    const size_t cap[2] = {1024, 16384};
    for (int ch = 0; ch < 2; ++ch) {
      klab_engine::Probe& pr = e->probe[ch];
      pr.a.assign(cap[ch], nullptr); pr.b.assign(cap[ch], nullptr); pr.flops.assign(cap[ch], 0.0);
      for (size_t i = 0; i < cap[ch]; ++i)
        if (hipEventCreate(&pr.a[i]) != hipSuccess || hipEventCreate(&pr.b[i]) != hipSuccess) return KLAB_ERR_UNSUPPORTED;
    }
  }
  e->probe_on = on != 0;
  if (on) for (auto& pr : e->probe) pr.n = 0;
  return 0;
}
extern "C" int klab_engine_probe_read(klab_engine* e, int channel, int* launches, float* total_ms, double* flops_total) {
  if (!e || channel < 0 || channel > 1) return KLAB_ERR_BADARG;
  klab_engine::Probe& pr = e->probe[channel];
  float tot = 0.f;
  double fl = 0;
  for (int i = 0; i < pr.n; ++i) {
    float ms = 0.f;
    hipError_t er = hipEventElapsedTime(&ms, pr.a[i], pr.b[i]);
    if (er != hipSuccess) return (int)er;
    tot += ms;
    fl += pr.flops[i];
  }
  if (launches) *launches = pr.n;
  if (total_ms) *total_ms = tot;
  if (flops_total) *flops_total = fl;
  return 0;
}
extern "C" const float* klab_engine_loss_ptr(const klab_engine* e) { return e ? e->loss : nullptr; }
// One-shot: the NEXT forward writes its mean loss to `out` (a device float the caller owns) instead of the engine's slot, so that a
// caller who must hand out a fresh tensor per step needs no copy launch behind the cross-entropy.  Ignored under graph replay
// (klab_engine_set_graph), where the loss stays at klab_engine_loss_ptr; returns 1 if it will be honoured, 0 if not.
extern "C" int klab_engine_set_loss_out(klab_engine* e, float* out) {
  if (!e) return KLAB_ERR_BADARG;
  e->loss_out = e->use_graph ? nullptr : out;
  return e->loss_out ? 1 : 0;
}
extern "C" const int* klab_engine_err_ptr(const klab_engine* e) { return e ? e->err_dev : nullptr; }
extern "C" const uint32_t* klab_engine_rng_ptr(const klab_engine* e) { return e ? e->seed_dev : nullptr; }

extern "C" const void* klab_engine_buffer(const klab_engine* e, const char* name, long* rows, long* cols, int* dtype) {
  if (!e || !e->bound || !name) return nullptr;
  const int d = e->cfg.main.d_model;
  auto set = [&](long r, long c2, int dt) { if (rows) *rows = r; if (cols) *cols = c2; if (dtype) *dtype = dt; };
  if (!strcmp(name, "encoder_input")) { set((long)e->B * e->Le, d, KLAB_F32); return e->enc.h[0]; }
  if (!strcmp(name, "encoder_out")) { set((long)e->B * e->Le, d, e->cfg.dtype); return e->enc.out_t; }
  if (!strcmp(name, "decoder_out")) { set((long)e->B * e->Lt, d, e->cfg.dtype); return e->dec.out_t; }
  if (!strcmp(name, "logits")) { set((long)e->B * e->Lt, e->cfg.main.vocab, e->cfg.dtype); return e->logits; }
  if (!strcmp(name, "logits_step")) { set((long)e->B, e->cfg.main.vocab, e->cfg.dtype); return e->dc_logits; }
  return nullptr;
}

// Greedy decoding with a K/V cache (ref/models/model.py:27-28 -> generate; HF/t5:308-332): the decoder over ONE new position
// t >= 1 per sample.  Precondition: a klab_engine_forward in evaluation mode on this binding (the prefill: encoder output,
// cross-attention K/V of all layers, the decoder's position-bias table, and -- position 0 being the start token whatever the
// target holds -- the self-attention K/V rows of position 0), then decode steps 1, 2, ... in order.  The per-layer q|k|v
// buffer [B*Lt, 3*inner] of the training path IS the cache: the new position's fused projection is written into row
// b*Lt + t, attention reads rows b*Lt + 0..t.  prev_tokens [B] = the ids generated at position t-1 (the decoder input at t,
// HF/t5:618-637).  Result: logits of position t in the "logits_step" buffer [B, vocab].
extern "C" int klab_engine_decode_step(klab_engine* e, int t, const long long* prev_tokens, void* stream) {
  if (!e || !e->bound || !prev_tokens || t < 1 || t >= e->Lt) return KLAB_ERR_BADARG;
  Ctx c{e, (hipStream_t)stream, e->cfg.dtype, e->es};
  const klab_t5_cfg& cfg = e->cfg.main;
  const auto& P = e->P[2];
  const auto& W = e->W[2];
  const int B = e->B, Lt = e->Lt, Le = e->Le, d = cfg.d_model, H = cfg.n_heads, dk = cfg.d_kv, inner = H * dk, ff = cfg.d_ff;
  const int nld = cfg.n_dec_layers;
  const long kv_ld = (long)nld * 2 * inner;
  RC(klab_embed_fwd(prev_tokens, 0, 1, cfg.start_id, cfg.pad_id, W[e->mi.shared], cfg.vocab, e->dc_h[0], B, d, 0.f, nullptr, 0, e->err_dev,
                    c.ws()));
  float* h = e->dc_h[0];
  float* h2 = e->dc_h[1];
  auto proj_res = [&](const void* x, int K, long woffv) -> int {  // h2 = h + x @ W^T  (f32 residual stream)
    klab_gemm_args g = G0(c, B, d, K, x, K, 1, woff(c, woffv), K, 1, h2, d, KLAB_F32);
    g.residual = h; g.ldr = d; g.r_dtype = KLAB_F32;
    RC(fwd_gemm(c, g, woffv));
    float* tmp = h; h = h2; h2 = tmp;
    return 0;
  };
  for (int i = 0; i < nld; ++i) {
    const T5LayerIdx& l = e->mi.dec[i];
    T5LayerBufs& b = e->dec.L[i];
    // self attention over the cache
    RC(klab_rmsnorm_fwd(h, W[l.ln0], e->dc_xn, c.dt, nullptr, e->dc_rstd, B, d, cfg.ln_eps, 0, 0, 0, 0.f, nullptr, 0, c.ws()));
    {
      klab_gemm_args g = G0(c, B, 3 * inner, d, e->dc_xn, d, 1, woff(c, P[l.q].warena_off), d, 1, eoff(c, b.qkv, (long)t * 3 * inner),
                            (long)Lt * 3 * inner, c.dt);
      RC(fwd_gemm(c, g, P[l.q].warena_off));
    }
    RC(klab_t5_decode_attn(c.dt, eoff(c, b.qkv, (long)t * 3 * inner), (long)Lt * 3 * inner, eoff(c, b.qkv, inner), eoff(c, b.qkv, 2 * inner),
                           (long)Lt * 3 * inner, 3 * inner, e->dec.bias + (long)t * Lt, (long)Lt * Lt, e->dc_ctx, inner, B, H, t + 1, dk,
                           c.ws()));
    RC(proj_res(e->dc_ctx, inner, P[l.o].warena_off));
    // cross attention over the encoder K/V projected at prefill
    RC(klab_rmsnorm_fwd(h, W[l.ln1], e->dc_xn, c.dt, nullptr, e->dc_rstd, B, d, cfg.ln_eps, 0, 0, 0, 0.f, nullptr, 0, c.ws()));
    RC(linear_fwd(c, e->dc_xn, B, d, P[l.cq].warena_off, inner, e->dc_q, inner, c.dt));
    RC(klab_t5_decode_attn(c.dt, e->dc_q, inner, eoff(c, e->kv_all, (long)i * 2 * inner), eoff(c, e->kv_all, (long)i * 2 * inner + inner),
                           (long)Le * kv_ld, kv_ld, nullptr, 0, e->dc_ctx, inner, B, H, Le, dk, c.ws()));
    RC(proj_res(e->dc_ctx, inner, P[l.co].warena_off));
    // feed forward
    RC(klab_rmsnorm_fwd(h, W[l.ln2], e->dc_xn, c.dt, nullptr, e->dc_rstd, B, d, cfg.ln_eps, 0, 0, 0, 0.f, nullptr, 0, c.ws()));
    RC(linear_fwd(c, e->dc_xn, B, d, P[l.wi].warena_off, ff, e->dc_hmid, ff, c.dt, nullptr, KLAB_ACT_RELU));
    RC(proj_res(e->dc_hmid, ff, P[l.wo].warena_off));
  }
  RC(klab_rmsnorm_fwd(h, W[e->mi.dec_final], e->dc_out, c.dt, nullptr, e->dc_rstd, B, d, cfg.ln_eps, 0, 0, 0, 0.f, nullptr, 0, c.ws()));
  {
    const int V = cfg.vocab;
    klab_gemm_args g = G0(c, B, V, d, e->dc_out, d, 1, woff(c, P[e->mi.shared].warena_off), d, 1, e->dc_logits, V, c.dt);
    g.alpha = cfg.scale_decoder_outputs ? 1.f / sqrtf((float)d) : 1.f;  // HF/t5:1044-1045
    RC(fwd_gemm(c, g, P[e->mi.shared].warena_off));
  }
  return 0;
}

// segment 0: LM head + decoder + shared embedding; 1: encoder; 2: Swin
static int backward_segment(klab_engine* e, int segment, const float* dloss_dev, void* stream) {
  Ctx c{e, (hipStream_t)stream, e->cfg.dtype, e->es};
  const klab_model_cfg& cfg = e->cfg;
  const int B = e->B, d = cfg.main.d_model, inner = cfg.main.n_heads * cfg.main.d_kv, nld = cfg.main.n_dec_layers;
  const float p = e->p_train;
  float* Gm = e->G[2];
  e->bucket_ev_live[segment] = false;
  if (segment == 0) {
    e->ev_next = 0;
    const int Md = B * e->Lt, V = cfg.main.vocab;
    const float alpha = cfg.main.scale_decoder_outputs ? 1.f / sqrtf((float)d) : 1.f;
    Ctx cs{e, e->side, c.dt, c.es};
    // Clearing 242 MB of gradient slices and the bias accumulators took 36 + 12 us of the main chain (four fills in front of the
    // kernels that need none of them).  They now run on the side stream, beside the LM-head input gradient; KLAB_ZERO_ON_SIDE=0:
    // the old order.
    static const bool zero_side = [] { const char* v = getenv("KLAB_ZERO_ON_SIDE"); return !v || atoi(v) != 0; }();
    const bool zside = zero_side && !e->use_graph;  // (host-side flags below: not for a captured sequence)
    if (zside) {
      RC(side_after_main(c));  // behind every earlier reader of the gradient buffers on the caller's stream (optimizer, running sums)
      RC((int)hipMemsetAsync(Gm + e->seg_off[0], 0, (size_t)e->seg_len[0] * 4, e->side));
      RC((int)hipMemsetAsync(Gm + e->seg_off[1], 0, (size_t)e->seg_len[1] * 4, e->side));
      RC((int)hipMemsetAsync(e->dec.dbias, 0, (size_t)cfg.main.n_heads * e->Lt * e->Lt * 4, e->side));
      RC((int)hipMemsetAsync(e->enc.dbias, 0, (size_t)cfg.main.n_heads * e->Le * e->Le * 4, e->side));
      RC((int)hipEventRecord(e->ev_zero, e->side));
      e->seg1_zeroed = e->dbias_zeroed[0] = e->dbias_zeroed[1] = true;
    } else {
      RC((int)hipMemsetAsync(Gm + e->seg_off[0], 0, (size_t)e->seg_len[0] * 4, c.s));
    }
    {  // d(dec_out) [Md,d] = dlogits [Md,V] @ shared [V,d]: K = vocabulary over only Md*d outputs => split-K, f32 atomics
      RC((int)hipMemsetAsync(e->dxn, 0, (size_t)Md * d * 4, c.s));
      klab_gemm_args g = G0(c, Md, d, V, e->logits, V, 1, woff(c, e->P[2][e->mi.shared].warena_off), d, 0, e->dxn, d, KLAB_F32);
      g.alpha = alpha; g.alpha_dev = dloss_dev; g.accumulate = 1; g.atomic_ok = 1;
      RC(klab_gemm(&g, c.ws()));
    }
    if (zside) RC((int)hipStreamWaitEvent(c.s, e->ev_zero, 0));  // (long since signalled: the fills take a fifth of that GEMM's time)
    // the weight gradient trails BEHIND the dgrad on the side stream (run side by side the two chip-filling GEMMs took
    // longer than one after the other); it then overlaps the decoder's first, latency-bound backward kernels
    RC(side_after_main(c));
    {  // d shared [V,d] = dlogits^T @ dec_out  (first of the tied weight's three contributors)
      klab_gemm_args g = G0(cs, V, d, Md, e->logits, V, 0, e->dec.out_t, d, 0, Gm + e->P[2][e->mi.shared].grad_off, d, KLAB_F32);
      g.alpha = alpha; g.alpha_dev = dloss_dev; g.accumulate = 1; g.atomic_ok = 1;
      // KLAB_LMHEAD_WGRAD_P8=1 (experiment): 256 x 256 tiles (252 of them, half the operand bytes of the 1004 tiles of 128 x 128)
      static const bool wg_p8 = [] { const char* v = getenv("KLAB_LMHEAD_WGRAD_P8"); return v && atoi(v) != 0; }();
      if (wg_p8) g.name_tag = 2;
      RC(klab_gemm(&g, cs.ws()));
    }
    float* dh0 = nullptr;
    RC(t5_stack_backward(c, cfg.main, e->P[2], e->W[2], Gm, e->mi.dec, e->mi.dec_final, e->dec, true, STACK_DEC, p, B, e->kv_all, e->dkv_all,
                         e->Le, (long)nld * 2 * inner, &dh0));
    // decoder input embedding: scatter-add into the tied table (second contributor); on the side stream BEHIND the
    // LM-head weight gradient, which read-modify-writes the same rows non-atomically
    RC(side_after_main(c));
    RC(klab_embed_bwd(e->last_tgt, 1, e->Lt, cfg.main.start_id, cfg.main.pad_id, dh0, Gm + e->P[2][e->mi.shared].grad_off, V, Md, d, p,
                      e->seed_dev, tag_of(STACK_DEC, 0, SITE_IN), cs.ws()));
    // cross-attention K/V projections of all layers at once: weights (side) + d(encoder output) (main)
    const int Me = B * e->Le, Nkv = nld * 2 * inner;
    if (e->kv_wgrad_done) e->kv_wgrad_done = false;  // released layer by layer with the decoder's groups (t5_stack_backward)
    else RC(linear_wgrad(cs, e->dkv_all, Nkv, e->enc.out_t, d, Me, Nkv, d, Gm + e->kvall_g_off));
    // d(encoder output) goes straight to dxn, where segment 1's stack reads d(final-norm output): no copy at the segment boundary
    e->denc_in_dxn = !e->use_graph;
    RC(linear_dgrad(c, e->dkv_all, Nkv, Me, Nkv, e->kvall_w_off, d, e->denc_in_dxn ? e->dxn : e->denc, KLAB_F32));
    return main_after_side(c);  // the segment's gradient slice is final for the caller's stream
  }
  if (segment == 1) {
    e->ev_next = 0;
    if (e->seg1_zeroed) e->seg1_zeroed = false;  // cleared at the entry of segment 0, beside the LM-head input gradient
    else RC((int)hipMemsetAsync(Gm + e->seg_off[1], 0, (size_t)e->seg_len[1] * 4, c.s));
    // the stack consumes dxn as d(final-norm output)
    const int Me = B * e->Le;
    if (e->denc_in_dxn) e->denc_in_dxn = false;
    else RC((int)hipMemcpyAsync(e->dxn, e->denc, (size_t)Me * d * 4, hipMemcpyDeviceToDevice, c.s));
    float* dh0 = nullptr;
    RC(t5_stack_backward(c, cfg.main, e->P[2], e->W[2], Gm, e->mi.enc, e->mi.enc_final, e->enc, false, STACK_ENC, p, B, nullptr, nullptr, 0, 0,
                         &dh0));
    if (cfg.train_swin) RC((int)hipMemcpyAsync(e->denc, dh0, (size_t)Me * d * 4, hipMemcpyDeviceToDevice, c.s));
    return main_after_side(c);
  }
  if (segment == 2) {
    if (!cfg.train_swin) return 0;
    return swin_backward(c, e->denc, p);
  }
  return KLAB_ERR_BADARG;
}


extern "C" int klab_engine_backward(klab_engine* e, int segment, const float* dloss_dev, void* stream) {
  if (!e || !e->bound || segment < 0 || segment > 2) return KLAB_ERR_BADARG;
  int erc = 0;
  const hipStream_t s = enter_stream(e, (hipStream_t)stream, erc);
  if (erc) return erc;
  const float* dl = e->dloss_buf;
  if (segment == 0) {  // d(objective)/d(loss): 1.0 when the caller passes NULL; replayed graphs read it at a fixed address
    if (dloss_dev && !e->use_graph) dl = dloss_dev;  // eager launches read the caller's scalar in place (stream-ordered)
    else if (dloss_dev) RC((int)hipMemcpyAsync(e->dloss_buf, dloss_dev, 4, hipMemcpyDeviceToDevice, s));
    else { static const float one = 1.f; RC((int)hipMemcpyAsync(e->dloss_buf, &one, 4, hipMemcpyHostToDevice, s)); }
  }
  RC(run_graphed(e, 4 + segment, s, [&]() { return backward_segment(e, segment, dl, (void*)s); }));
  return leave_stream(e, (hipStream_t)stream, s);
}

namespace {

// Swin-V2 backward (only with --image_model_train; Swin itself is always in eval mode).  dh0 is
// d loss / d (encoder input) [B*Le, d] in f32; rows [0, N_img) of each image belong to Swin.
int swin_backward(const Ctx& c, const float* dh0, float p_in) {
  klab_engine* e = c.e;
  const klab_swin_cfg& s = e->cfg.swin;
  const auto& P = e->P[0];
  const auto& W = e->W[0];
  float* Gs = e->G[0];
  auto G = [&](int pi) { return Gs + P[pi].grad_off; };
  const int B = e->B, R0 = s.image_size / s.patch, C0 = s.embed_dim;
  RC((int)hipMemsetAsync(Gs + e->seg_off[2], 0, (size_t)e->seg_len[2] * 4, c.s));
  RC((int)hipMemsetAsync(e->sdbias, 0, e->sdz_bytes, c.s));  // every block's bias-gradient slices at once
  // final LN backward: input x = last hidden (f32), dout in the remapped encoder-input rows, with the input dropout
  const int last = s.n_stages - 1;
  const int Cl = C0 << last;
  const SwinStageBufs& lsb = e->sw[last];
  const float* xlast = lsb.blk.empty() ? (last > 0 ? e->sw[last - 1].xm : e->x0) : lsb.blk.back().h2;
  float* dh = e->sdh_a;   // d loss / d hidden stream (f32)
  float* dh2 = e->sdh_b;
  RC(klab_layernorm_bwd(dh0, xlast, KLAB_F32, W[e->si.lnw], e->sw_fmean, e->sw_frstd, dh, G(e->si.lnw), G(e->si.lnb), B * e->Nimg, Cl,
                        e->Nimg, e->Le, 0, p_in, e->seed_dev, tag_of(STACK_ENC, 0, SITE_IN), c.ws()));
  long bias_off_total = 0;
  for (int st = 0; st < s.n_stages; ++st) bias_off_total += 3L * (C0 << st) * s.depths[st];
  long bias_off = bias_off_total;
  (void)bias_off;
  int swin_bucket = 0, blk_no = 0;
  static const bool side_env = [] { const char* v = getenv("KLAB_SWIN_SIDE_WGRAD"); return !v || atoi(v) != 0; }();
  const bool side_on = side_env && e->sdyA[0] != nullptr;
  const Ctx cs{e, e->side, c.dt, c.es};
  for (int st = last; st >= 0; --st) {
    SwinStageBufs& sb = e->sw[st];
    const int R = R0 >> st, C = C0 << st;
    if (st < last) {
      // patch merging backward: LN -> reduction -> gather
      const long M2 = (long)B * (R / 2) * (R / 2);
      RC(klab_layernorm_bwd(dh, sb.mo, c.dt, W[e->si.st[st].mnw], sb.mmean, sb.mrstd, e->sdy, G(e->si.st[st].mnw), G(e->si.st[st].mnb), (int)M2,
                            2 * C, 0, 0, 0, 0.f, nullptr, 0, c.ws()));
      RC(linear_wgrad(c, e->sdy, 2 * C, sb.mg, 4 * C, (int)M2, 2 * C, 4 * C, G(e->si.st[st].redw)));
      RC(linear_dgrad(c, e->sdy, 2 * C, (int)M2, 2 * C, P[e->si.st[st].redw].warena_off, 4 * C, e->sdm, KLAB_F32));
      RC(klab_merge_scatter(e->sdm, dh2, B, R, C, c.ws()));
      float* t = dh; dh = dh2; dh2 = t;
    }
    for (int k = (int)sb.blk.size() - 1; k >= 0; --k) {
      SwinBlockBufs& q = sb.blk[k];
      const SwinBlockIdx& ix = e->si.st[st].blk[k];
      const int M = (int)q.M, F = s.mlp_ratio * C, n = q.w * q.w;
      // The activation-gradient chain (LayerNorm', dgrad GEMMs, attention backward) stays on the main stream; the block's
      // bias column sums and weight-gradient GEMMs trail on the side stream (as the T5 stacks do).  Their operands live in
      // one of two alternating buffer sets: before the main stream rewrites a set it waits for the side stream's event of the
      // block that used it last.
      const int par = blk_no & 1;
      if (side_on && blk_no >= 2) RC((int)hipStreamWaitEvent(c.s, e->swin_done_ev[par], 0));
      void* dyA = side_on ? e->sdyA[par] : e->sdy;     // d fo   (LayerNorm-after backward)
      void* dyB = side_on ? e->sdyB[par] : e->sdy;     // d po   (LayerNorm-before backward)
      void* da = side_on ? e->sdaP[par] : e->sda;      // d z
      void* dqkv = side_on ? e->sdqkvP[par] : e->sdqkv;
      const Ctx& cw = side_on ? cs : c;                // where the weight gradients go
      // h2 = h1 + LN2(fo):  d fo = LN2'(dh);  dh flows through the shortcut unchanged
      // (the column sums of d fo are fc2's bias gradient: folded into the same pass)
      RC(klab_layernorm_bwd_bias(dh, q.fo, c.dt, W[ix.ln2w], q.mean2, q.rstd2, dyA, G(ix.ln2w), G(ix.ln2b), G(ix.f2b), M, C, 0, 0, 0, 0.f, nullptr,
                                 0, c.ws()));
      if (!side_on) RC(linear_wgrad(c, dyA, C, q.a, F, M, C, F, G(ix.f2w)));
      {  // d z = (d fo @ W2) * gelu'(z)
        klab_gemm_args g = G0(c, M, F, C, dyA, C, 1, woff(c, P[ix.f2w].warena_off), F, 0, da, F, c.dt);
        g.aux = q.z; g.ldaux = F; g.aux_mode = KLAB_AUX_DGELU;
        RC(klab_gemm(&g, c.ws()));
      }
      if (!side_on) {
        RC(klab_colsum(da, F, c.dt, M, F, G(ix.f1b), c.ws()));
        RC(linear_wgrad(c, da, F, q.h1t, C, M, F, C, G(ix.f1w)));
      }
      {  // dh1 = dh + d z @ W1   (accumulate into the stream gradient)
        klab_gemm_args g = G0(c, M, C, F, da, F, 1, woff(c, P[ix.f1w].warena_off), C, 0, dh, C, KLAB_F32);
        g.accumulate = 1;
        RC(klab_gemm(&g, c.ws()));
      }
      // h1 = x + LN1(po)
      RC(klab_layernorm_bwd_bias(dh, q.po, c.dt, W[ix.ln1w], q.mean1, q.rstd1, dyB, G(ix.ln1w), G(ix.ln1b), G(ix.pb), M, C, 0, 0, 0, 0.f, nullptr,
                                 0, c.ws()));
      if (!side_on) RC(linear_wgrad(c, dyB, C, q.ctx, C, M, C, C, G(ix.pw)));
      RC(linear_dgrad(c, dyB, C, M, C, P[ix.pw].warena_off, C, e->sdctx, c.dt));
      klab_swin_attn_args a;
      memset(&a, 0, sizeof(a));
      a.dtype = c.dt; a.qkv = q.qkv; a.ctx = q.ctx; a.bias = q.bias; a.bias_table = q.btab; a.logit_scale = W[ix.ls]; a.lse = q.lse;
      a.B = B; a.R = q.R; a.w = q.w; a.shift = q.shift; a.H = q.H; a.C = C;
      a.dctx = e->sdctx; a.dqkv = dqkv; a.dlogit_scale = G(ix.ls);
      a.bwd_ws = e->sattn_ws; a.bwd_ws_bytes = e->sattn_ws_bytes;
      if (ix.vb >= 0) { a.v_bias = e->farena + P[ix.vb].farena_off; a.dv_bias = G(ix.vb); }  // padded windows
      float* blk_dbias = e->sdbias + blk_no * e->sdb_stride;   // this block's slices: clear since the head of swin_backward
      float* blk_dbtab = e->sdbtab + blk_no * e->sdt_stride;
      float* blk_dtable = e->sdtable + blk_no * e->sdt_stride;
      if (q.btab) {  // large window: the bias gradient is accumulated per table entry
        a.dbias_table = blk_dbtab;
        RC(klab_swin_attn_bwd(&a, c.ws()));
        RC(klab_swin_cpb_table_bwd(blk_dbtab, q.btab, e->swin_coords[st], q.hidden, W[ix.c2w], blk_dtable, G(ix.c0w), G(ix.c0b), G(ix.c2w),
                                   e->swin_ntab[st], q.H, 512, c.ws()));
      } else {
        a.dbias = blk_dbias;
        RC(klab_swin_attn_bwd(&a, c.ws()));
        RC(klab_swin_cpb_bias_bwd_pz(blk_dbias, q.bias, e->swin_index[st], e->swin_coords[st], q.hidden, W[ix.c0w], W[ix.c2w], blk_dtable,
                                     G(ix.c0w), G(ix.c0b), G(ix.c2w), e->swin_ntab[st], n, q.H, 512, 1, c.ws()));
      }
      if (side_on) {  // everything the block's weight gradients read exists now: release them to the side stream
        RC(side_after_main(c));
        RC(klab_colsum(da, F, c.dt, M, F, G(ix.f1b), cw.ws()));
      }
      if (ix.qb >= 0) {
        RC(klab_colsum(dqkv, 3 * C, c.dt, M, C, G(ix.qb), cw.ws()));
        RC(klab_colsum((char*)dqkv + (size_t)2 * C * c.es, 3 * C, c.dt, M, C, G(ix.vb), cw.ws()));
      }
      if (side_on) {  // the block's four weight gradients in ONE grouped launch (members outside its form run one by one)
        klab_gemm_args gs[4];
        auto wg = [&](int i, const void* dy, long lddy, const void* x, long ldx, int N, int K, float* dw) {
          gs[i] = G0(cw, N, K, M, dy, lddy, 0, x, ldx, 0, dw, K, KLAB_F32);  // as linear_wgrad
          gs[i].accumulate = 1; gs[i].atomic_ok = 1;
        };
        wg(0, dyA, C, q.a, F, C, F, G(ix.f2w));
        wg(1, da, F, q.h1t, C, F, C, G(ix.f1w));
        wg(2, dyB, C, q.ctx, C, C, C, G(ix.pw));
        wg(3, dqkv, 3 * C, q.xt_in, C, 3 * C, C, G(ix.qw));  // q|k|v grads adjacent
        RC(klab_gemm_grouped(gs, 4, cw.ws()));
      } else {
        RC(linear_wgrad(cw, dqkv, 3 * C, q.xt_in, C, M, 3 * C, C, G(ix.qw)));  // q|k|v grads adjacent
      }
      if (e->bucket_events_on && !e->use_graph && swin_bucket < (int)e->bucket_ev[2].size())  // this block's GEMM-weight gradients are final
        RC((int)hipEventRecord(e->bucket_ev[2][swin_bucket], cw.s));
      ++swin_bucket;
      if (side_on) RC((int)hipEventRecord(e->swin_done_ev[par], cs.s));
      ++blk_no;
      {
        klab_gemm_args g = G0(c, M, C, 3 * C, dqkv, 3 * C, 1, woff(c, P[ix.qw].warena_off), C, 0, dh, C, KLAB_F32);
        g.accumulate = 1;
        RC(klab_gemm(&g, c.ws()));
      }
    }
  }
  if (side_on) RC(main_after_side(c));  // every weight gradient of the tower is behind this point of the caller's stream
  // patch embedding: LN -> conv-as-GEMM (weights + bias only; pixels need no gradient)
  const long M0 = (long)B * R0 * R0;
  const int K0 = s.in_ch * s.patch * s.patch;
  RC(klab_layernorm_bwd_bias(dh, e->pe_out, c.dt, W[e->si.penw], e->pe_mean, e->pe_rstd, e->sdy, G(e->si.penw), G(e->si.penb), G(e->si.peb), (int)M0,
                             C0, 0, 0, 0, 0.f, nullptr, 0, c.ws()));
  RC(linear_wgrad(c, e->sdy, C0, e->cols, e->pe_kp, (int)M0, C0, K0, G(e->si.pew)));
  e->bucket_ev_live[2] = e->bucket_events_on && !e->use_graph;
  return 0;
}

}  // namespace
