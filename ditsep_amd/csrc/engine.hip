// Engine + C-ABI (include/ditsep_hip.h) of the MI355X separation path.
//
// Host side of the native library: owns the packed weights and the activation
// workspace in HBM, and turns one call of the reference's entry points into a
// fixed sequence of gfx950 kernel launches on the caller's HIP stream
// (optionally captured into a hipGraph and replayed).  Nothing here touches
// torch; the Python facade passes raw device pointers.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ditsep_hip.h"
#include "igemm.h"
#include "kernels.h"

namespace {

struct Err : std::runtime_error {
  int code;
  Err(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] void fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Err(code, buf);
}

#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) fail(DSN_EHIP, "%s -> %s (%s:%d)", #expr, hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                        \
  } while (0)

struct DevTensor {
  float* p = nullptr;
  std::vector<long> shape;
  long numel = 0;
};

struct Packed {  // K-major packed operand planes of one contraction
  op16_t* w = nullptr;
  long ps = 0;
  int N = 0, K = 0, Cin = 0, taps = 0;
  float* bias = nullptr;
  int bias_mod = 1;
};

struct ActP {  // producer-side activation
  int kind = DSN_ACT_NONE;
  float* a = nullptr;
  float* ib = nullptr;
  int mod = 1;
};

struct Packed8 {  // fp8 (MX) packed Linear weight: e4m3 bytes [N][K] + E8M0 block scales [N][K/32]
  unsigned char* w = nullptr;
  unsigned char* s = nullptr;
  int N = 0, K = 0;
  float* bias = nullptr;
};

struct DitLayer {
  float *g1 = nullptr, *be1 = nullptr, *g2 = nullptr, *be2 = nullptr;
  Packed qkv, out, ff1, ff2;
  Packed8 qkv8, out8, ff1_8, ff2_8;  // DSN_PREC_FP8
  // ff_norm folded into FF-in (single-plane modes): W * diag(gamma) packed, its rounded row sums, bias + W beta
  Packed ff1f;
  float* ff1_colsum = nullptr;
  Packed8 ff1f8;  // fp8 mode: the same fold on the MX-quantised W * diag(gamma), row sums of the dequantised bytes
  float* ff1_colsum8 = nullptr;
};

struct ResUnit {
  ActP act0, act2;  // act0 is applied by the PRODUCER of the unit's input
  Packed conv7, conv1;
  int dil = 1;
};

struct VaeBlock {
  ActP act;     // block-level activation (before convT / before strided conv)
  Packed conv;  // ConvTranspose1d (decoder) or strided Conv1d (encoder)
  ResUnit ru[3];
  int cin = 0, cout = 0, stride = 1;
};

struct Graph {
  hipGraphExec_t exec = nullptr;
  uint64_t epoch = ~0ull;  // workspace epoch the graph (or its eager warm-up) was built against
  bool warmed = false;
};

struct ProfRec {
  hipEvent_t a, b;
  double flops;
  double hbm_bytes = 0;  // algorithmic HBM bytes of the launch (0 = not stated)
  const char* tag = "";  // call-site label (static string): rows of dsn_profile_rows
  bool gemm = true;      // counted in the implicit-GEMM family totals of dsn_profile_end
  bool hbm_bound = false;  // the HBM-bound member of that family (fused ResidualUnit): dsn_profile_hbm
};

struct ProfRow {
  std::string name;
  double ms = 0, flops = 0, bytes = 0;
  int64_t launches = 0;
};

}  // namespace

struct dsn_ctx {
  dsn_config cfg;
  std::string err;
  int P = 2;   // operand planes
  int PL = 2;  // DSN_PL(P, fp16 flag) as the kernels take it
  bool fp8 = false;  // DSN_PREC_FP8: DiT layer GEMMs on fp8 (MX) operands, the rest single-plane fp16
  bool fold_ln8 = false;  // the same in the fp8 mode (raw x' as e4m3 + block scales)
  bool fold_ln = false;  // ff_norm folded into FF-in, to_out without split-K writing the residual stream itself
  bool finalized = false;
  bool use_graphs = false;
  // producer-finished GroupNorm (GemmDesc::gnf_out): host-mapped give-up flag of its waits; counters per (row tiles per
  // image) value, zeroed once -- they only grow
  int* fin_err_host = nullptr;
  int* fin_err_dev = nullptr;
  std::map<int, std::pair<unsigned*, long>> gnf_sync;
  unsigned* gnf_counters(int nper, long count, hipStream_t st) {
    if (!fin_err_host) {
      HIPCHK(hipHostMalloc((void**)&fin_err_host, sizeof(int), hipHostMallocMapped));
      *fin_err_host = 0;
      HIPCHK(hipHostGetDevicePointer((void**)&fin_err_dev, fin_err_host, 0));
    }
    auto& c = gnf_sync[nper];
    if (!c.first || c.second < count) {
      if (c.first) {
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipFree(c.first));
      }
      c.second = std::max(count, 1024L);
      HIPCHK(hipMalloc((void**)&c.first, c.second * sizeof(unsigned)));
      HIPCHK(hipMemsetAsync(c.first, 0, c.second * sizeof(unsigned), st));
      ++ws_epoch;  // (first use is an eager warm-up call, never a capture)
    }
    return c.first;
  }
  double hbm_ms = 0, hbm_bytes = 0;  // HBM-bound launches of the last profiled region (dsn_profile_hbm)
  int64_t hbm_launches = 0;
  std::map<std::string, DevTensor> raw;
  std::vector<void*> allocs;
  std::map<std::string, std::pair<void*, size_t>> ws;
  uint64_t ws_epoch = 0;  // bumped on every workspace (re)allocation -> graphs invalid

  // DiT
  float* tf_w = nullptr;
  Packed t1, t2, pin, pout;  // pin / pout carry the folded pre/postprocess convs (fold_dit_io)
  std::vector<DitLayer> layers;
  // VAE
  Packed dec_in;
  std::vector<VaeBlock> dec_blocks;
  ActP dec_final_act;
  float* dec_out_w = nullptr;  // [7][C0]
  int dec_out_taps = 7;
  float *enc_in_w = nullptr, *enc_in_b = nullptr;
  std::vector<VaeBlock> enc_blocks;
  ActP enc_final_act;
  Packed enc_out;

  std::map<std::string, Graph> graphs;
  // capture / replay streams (graphs cannot be captured on the NULL stream): one per entry-point family,
  // so that the decode of one batch can overlap the sampler of the next when the caller issues them on
  // different streams (bench.py pipelines them)
  hipStream_t own[2] = {nullptr, nullptr};
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
  bool profiling = false;
  std::vector<ProfRec> prof;
  std::vector<ProfRow> prof_rows;  // per call-site aggregation of the last profiled region
  const char* cur_tag = "";        // label the next profiled launches carry
  struct Tag {                     // scoped call-site label
    dsn_ctx* c;
    const char* prev;
    Tag(dsn_ctx* ctx, const char* t) : c(ctx), prev(ctx->cur_tag) { c->cur_tag = t; }
    ~Tag() { c->cur_tag = prev; }
  };
  // bracket a non-GEMM launch with events when profiling (algorithmic HBM bytes given by the caller)
  template <class F>
  void prof_launch(const char* tag, double bytes, hipStream_t st, F&& f) {
    if (!profiling) {
      f();
      return;
    }
    ProfRec pr;
    HIPCHK(hipEventCreate(&pr.a));
    HIPCHK(hipEventCreate(&pr.b));
    pr.flops = 0;
    pr.hbm_bytes = bytes;
    pr.tag = tag;
    pr.gemm = false;
    HIPCHK(hipEventRecord(pr.a, st));
    f();
    HIPCHK(hipEventRecord(pr.b, st));
    prof.push_back(pr);
  }
  std::vector<float> tv_host;  // uploaded timestep table signature
  int tv_B = -1;

  // Run `body(stream)` -- a pure sequence of kernel launches over workspace pointers --
  // either eagerly or as a cached hipGraph.  The first call with a given key runs
  // eagerly (it may grow the workspace), the second captures, later ones replay.
  template <class F>
  void run_graphed(const std::string& key, hipStream_t st, F&& body) {
    if (!use_graphs || profiling) {
      body(st);
      return;
    }
    Graph& g = graphs[key];
    if (g.exec && g.epoch == ws_epoch) {
      HIPCHK(hipGraphLaunch(g.exec, st));
      return;
    }
    if (g.exec) {
      (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
    }
    if (!g.warmed || g.epoch != ws_epoch) {
      body(st);
      g.warmed = true;
      g.epoch = ws_epoch;
      return;
    }
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    try {
      body(st);
    } catch (...) {
      (void)hipStreamEndCapture(st, &graph);
      if (graph) (void)hipGraphDestroy(graph);
      throw;
    }
    HIPCHK(hipStreamEndCapture(st, &graph));
    hipError_t e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) fail(DSN_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    HIPCHK(hipGraphLaunch(g.exec, st));
  }
  // stream the engine enqueues on: the caller's in eager mode, its own (ordered after the
  // caller's by an event) in graph mode
  hipStream_t enter(hipStream_t caller, int which = 0) {
    if (!use_graphs || profiling) return caller;
    if (!own[which]) {
      HIPCHK(hipStreamCreateWithFlags(&own[which], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&ev_in[which], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&ev_out[which], hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(ev_in[which], caller));
    HIPCHK(hipStreamWaitEvent(own[which], ev_in[which], 0));
    return own[which];
  }
  void leave(hipStream_t caller, hipStream_t used, int which = 0) {
    if (used == caller) return;
    HIPCHK(hipEventRecord(ev_out[which], used));
    HIPCHK(hipStreamWaitEvent(caller, ev_out[which], 0));
  }

  // ---------------------------------------------------------------- memory
  void* dmalloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) fail(DSN_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    allocs.push_back(p);
    return p;
  }
  template <class T>
  T* wsbuf(const std::string& name, size_t count) {
    const size_t bytes = count * sizeof(T) + 256;
    auto it = ws.find(name);
    if (it != ws.end() && it->second.second >= bytes) return (T*)it->second.first;
    if (it != ws.end()) {
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipFree(it->second.first));
      ws.erase(it);
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) fail(DSN_ENOMEM, "workspace %s: hipMalloc(%zu) failed", name.c_str(), bytes);
    ws[name] = {p, bytes};
    ++ws_epoch;
    return (T*)p;
  }
  int64_t ws_bytes() const {
    int64_t t = 0;
    for (auto& kv : ws) t += (int64_t)kv.second.second;
    return t;
  }

  // ---------------------------------------------------------------- weights
  mutable std::set<std::string> used;  // names finalize() consumed (strict loading: the rest is unexpected)
  const DevTensor& get(const std::string& name) const {
    auto it = raw.find(name);
    if (it == raw.end()) fail(DSN_ESTATE, "missing weight '%s'", name.c_str());
    used.insert(name);
    return it->second;
  }
  bool has(const std::string& name) const { return raw.count(name) != 0; }
  float* maybe(const std::string& name) const { return has(name) ? get(name).p : nullptr; }

  std::vector<float> to_host(const std::string& name) const {
    const DevTensor& t = get(name);
    std::vector<float> h((size_t)t.numel);
    HIPCHK(hipMemcpy(h.data(), t.p, sizeof(float) * h.size(), hipMemcpyDeviceToHost));
    return h;
  }
  void set_raw(const std::string& name, const std::vector<float>& h, std::vector<long> shape) {
    auto it = raw.find(name);
    if (it != raw.end()) {
      (void)hipFree(it->second.p);
      raw.erase(it);
    }
    DevTensor t;
    t.shape = std::move(shape);
    t.numel = (long)h.size();
    HIPCHK(hipMalloc((void**)&t.p, sizeof(float) * h.size()));
    HIPCHK(hipMemcpy(t.p, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
    raw[name] = t;
  }
  // The DiT wraps its transformer in two residual 1x1 convs without bias (dit.py: preprocess_conv before
  // project_in, postprocess_conv after project_out).  Both pairs are linear, so they are folded once (fp64 on
  // the host) into one matrix each:  Win (I + Wpre)  and  (I + Wpost) Wout  -- two small GEMMs less per call.
  void fold_dit_io(const std::string& sp) {
    const DevTensor& win = get(sp + "transformer.project_in.weight");
    const DevTensor& wout = get(sp + "transformer.project_out.weight");
    const long D = win.shape[0], din = win.numel / D, io = wout.shape[0];
    if (wout.numel != io * D || get(sp + "preprocess_conv.weight").numel != din * din ||
        get(sp + "postprocess_conv.weight").numel != io * io)
      fail(DSN_EINVAL, "DiT: preprocess/project_in/project_out/postprocess shapes do not chain");
    const std::vector<float> a = to_host(sp + "transformer.project_in.weight"), pre = to_host(sp + "preprocess_conv.weight");
    const std::vector<float> b = to_host(sp + "transformer.project_out.weight"), post = to_host(sp + "postprocess_conv.weight");
    std::vector<float> fin((size_t)(D * din)), fout((size_t)(io * D));
    for (long r = 0; r < D; ++r)
      for (long c = 0; c < din; ++c) {
        double acc = a[r * din + c];
        for (long k = 0; k < din; ++k) acc += (double)a[r * din + k] * (double)pre[k * din + c];
        fin[r * din + c] = (float)acc;
      }
    std::vector<double> row((size_t)D);
    for (long r = 0; r < io; ++r) {
      for (long c = 0; c < D; ++c) row[c] = b[r * D + c];
      for (long k = 0; k < io; ++k) {
        const double w = post[r * io + k];
        for (long c = 0; c < D; ++c) row[c] += w * (double)b[k * D + c];
      }
      for (long c = 0; c < D; ++c) fout[r * D + c] = (float)row[c];
    }
    set_raw(sp + "__project_in_folded", fin, {D, din});
    set_raw(sp + "__project_out_folded", fout, {io, D});
  }

  Packed pack_linear(const std::string& wname, const std::string& bname, bool swiglu, hipStream_t st) {
    const DevTensor& w = get(wname);
    if (w.shape.size() < 2) fail(DSN_EINVAL, "%s: expected a matrix", wname.c_str());
    Packed p;
    p.N = (int)w.shape[0];
    p.K = (int)(w.numel / w.shape[0]);
    p.Cin = p.K;
    p.taps = 1;
    p.ps = (long)p.N * p.K;
    p.w = (op16_t*)dmalloc(sizeof(op16_t) * p.ps * P);
    launch_pack_weight(w.p, nullptr, p.w, p.ps, PL, swiglu ? PACK_LINEAR_SWIGLU : PACK_LINEAR, p.N, p.K, p.K, p.N,
                       1, 1, st);
    if (!bname.empty() && has(bname)) {
      const DevTensor& b = get(bname);
      if (b.numel != p.N) fail(DSN_EINVAL, "%s: bias size mismatch", bname.c_str());
      if (swiglu) {
        p.bias = (float*)dmalloc(sizeof(float) * p.N);
        launch_pack_bias_swiglu(b.p, p.bias, p.N, st);
      } else {
        p.bias = b.p;
      }
      p.bias_mod = p.N;
    }
    return p;
  }

  Packed8 pack_linear_fp8(const std::string& wname, const std::string& bname, bool swiglu, hipStream_t st) {
    const DevTensor& w = get(wname);
    if (w.shape.size() < 2) fail(DSN_EINVAL, "%s: expected a matrix", wname.c_str());
    Packed8 p;
    p.N = (int)w.shape[0];
    p.K = (int)(w.numel / w.shape[0]);
    if (p.K % 128 != 0) fail(DSN_EINVAL, "%s: fp8 GEMMs need K %% 128 == 0 (got %d)", wname.c_str(), p.K);
    p.w = (unsigned char*)dmalloc((size_t)p.N * p.K);
    p.s = (unsigned char*)dmalloc((size_t)p.N * (p.K / 32));
    launch_pack_weight_fp8(w.p, p.w, p.s, p.N, p.K, swiglu ? 1 : 0, st);
    if (!bname.empty() && has(bname)) {
      const DevTensor& b = get(bname);
      if (b.numel != p.N) fail(DSN_EINVAL, "%s: bias size mismatch", bname.c_str());
      if (swiglu) {
        p.bias = (float*)dmalloc(sizeof(float) * p.N);
        launch_pack_bias_swiglu(b.p, p.bias, p.N, st);
      } else {
        p.bias = b.p;
      }
    }
    return p;
  }

  // weight-normed Conv1d [Cout][Cin][kw] (or plain `weight`) -> [Cout][kw*Cin]
  Packed pack_conv(const std::string& prefix, hipStream_t st) {
    const bool wn = has(prefix + "weight_v");
    const DevTensor& v = get(prefix + (wn ? "weight_v" : "weight"));
    if (v.shape.size() != 3) fail(DSN_EINVAL, "%s: expected [Cout,Cin,k]", prefix.c_str());
    const int Cout = (int)v.shape[0], Cin = (int)v.shape[1], kw = (int)v.shape[2];
    float* scale = nullptr;
    if (wn) {
      scale = (float*)dmalloc(sizeof(float) * Cout);
      launch_wn_scale(v.p, get(prefix + "weight_g").p, scale, Cout, (long)Cin * kw, st);
    }
    Packed p;
    p.N = Cout;
    p.Cin = Cin;
    p.taps = kw;
    p.K = Cin * kw;
    p.ps = (long)p.N * p.K;
    p.w = (op16_t*)dmalloc(sizeof(op16_t) * p.ps * P);
    launch_pack_weight(v.p, scale, p.w, p.ps, PL, PACK_CONV, p.N, p.K, Cin, Cout, kw, 1, st);
    p.bias = maybe(prefix + "bias");
    p.bias_mod = Cout;
    return p;
  }

  // weight-normed ConvTranspose1d [Cin][Cout][2s] -> phase GEMM [s*Cout][2*Cin]
  Packed pack_convT(const std::string& prefix, int stride, hipStream_t st) {
    const bool wn = has(prefix + "weight_v");
    const DevTensor& v = get(prefix + (wn ? "weight_v" : "weight"));
    if (v.shape.size() != 3) fail(DSN_EINVAL, "%s: expected [Cin,Cout,k]", prefix.c_str());
    const int Cin = (int)v.shape[0], Cout = (int)v.shape[1], kw = (int)v.shape[2];
    if (kw != 2 * stride) fail(DSN_EINVAL, "%s: kernel %d != 2*stride %d", prefix.c_str(), kw, stride);
    float* scale = nullptr;
    if (wn) {
      scale = (float*)dmalloc(sizeof(float) * Cin);
      launch_wn_scale(v.p, get(prefix + "weight_g").p, scale, Cin, (long)Cout * kw, st);
    }
    Packed p;
    p.N = stride * Cout;
    p.Cin = Cin;
    p.taps = 2;
    p.K = 2 * Cin;
    p.ps = (long)p.N * p.K;
    p.w = (op16_t*)dmalloc(sizeof(op16_t) * p.ps * P);
    launch_pack_weight(v.p, scale, p.w, p.ps, PL, PACK_CONVT, p.N, p.K, Cin, Cout, kw, stride, st);
    p.bias = maybe(prefix + "bias");
    p.bias_mod = Cout;
    return p;
  }

  ActP make_act(const std::string& prefix, int channels, hipStream_t st) {
    ActP a;
    a.mod = channels;
    if (!cfg.vae_use_snake) {
      a.kind = DSN_ACT_ELU;
      return a;
    }
    a.kind = DSN_ACT_SNAKE;
    a.a = (float*)dmalloc(sizeof(float) * channels);
    a.ib = (float*)dmalloc(sizeof(float) * channels);
    launch_snake_params(get(prefix + "alpha").p, get(prefix + "beta").p, a.a, a.ib, channels, st);
    return a;
  }

  ResUnit make_ru(const std::string& prefix, int ch, int dil, hipStream_t st) {
    ResUnit r;
    r.dil = dil;
    r.act0 = make_act(prefix + "layers.0.", ch, st);
    r.conv7 = pack_conv(prefix + "layers.1.", st);
    r.act2 = make_act(prefix + "layers.2.", ch, st);
    r.conv1 = pack_conv(prefix + "layers.3.", st);
    return r;
  }

  // Tensors the configured network does not consume.  Known buffers of the reference's modules are allowed
  // (RotaryEmbedding.inv_freq, BatchNorm counters); anything else under score_model.* / vae.* means the checkpoint
  // was trained as a different network (cross-attention, adaLN, global conditioning, qk-norm ...): strict mode
  // refuses it (nn.Module.load_state_dict(strict=True) semantics, diffsep_latent.py loads strictly), non-strict
  // drops it.  Either way the unused tensors do not stay resident in HBM.
  void check_unused(bool strict) {
    auto ends_with = [](const std::string& s, const char* suf) {
      const size_t n = strlen(suf);
      return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
    };
    std::vector<std::string> extra, drop;
    for (auto& kv : raw) {
      if (used.count(kv.first)) continue;
      drop.push_back(kv.first);
      if (ends_with(kv.first, "inv_freq") || ends_with(kv.first, "num_batches_tracked")) continue;
      extra.push_back(kv.first);
    }
    if (strict && !extra.empty()) {
      std::string msg;
      for (size_t i = 0; i < extra.size() && i < 6; ++i) msg += (i ? ", " : "") + extra[i];
      if (extra.size() > 6) msg += ", ... (+" + std::to_string(extra.size() - 6) + " more)";
      fail(DSN_ESTATE, "%zu unexpected tensor(s) the configured network does not use: %s", extra.size(), msg.c_str());
    }
    for (auto& name : drop) {
      (void)hipFree(raw[name].p);
      raw.erase(name);
    }
  }

  void finalize(hipStream_t st, bool strict = true) {
    used.clear();
    if (!allocs.empty()) {  // weights replaced (e.g. EMA <-> raw parameters): drop the old packing, graphs are stale
      HIPCHK(hipDeviceSynchronize());
      for (void* q : allocs) (void)hipFree(q);
      allocs.clear();
      ++ws_epoch;
    }
    const int D = cfg.dit_embed_dim;
    fold_ln = P == 1 && !fp8 && D % 64 == 0 && getenv("DSN_NO_LN_FOLD") == nullptr;
    fold_ln8 = fp8 && D % 128 == 0 && getenv("DSN_NO_LN_FOLD") == nullptr;
    if (cfg.score_kind == DSN_SCORE_DIT) {
      const std::string sp = "score_model.";
      if (D % 64 != 0 || cfg.dit_heads <= 0 || D / cfg.dit_heads != 64 || D % cfg.dit_heads != 0)
        fail(DSN_EINVAL, "DiT: embed_dim must be heads x 64 (got %d / %d heads)", D, cfg.dit_heads);
      tf_w = get(sp + "timestep_features.weight").p;
      t1 = pack_linear(sp + "to_timestep_embed.0.weight", sp + "to_timestep_embed.0.bias", false, st);
      t2 = pack_linear(sp + "to_timestep_embed.2.weight", sp + "to_timestep_embed.2.bias", false, st);
      fold_dit_io(sp);
      pin = pack_linear(sp + "__project_in_folded", "", false, st);
      pout = pack_linear(sp + "__project_out_folded", "", false, st);
      layers.resize(cfg.dit_depth);
      for (int i = 0; i < cfg.dit_depth; ++i) {
        const std::string lp = sp + "transformer.layers." + std::to_string(i) + ".";
        DitLayer& L = layers[i];
        L.g1 = get(lp + "pre_norm.gamma").p;
        L.be1 = maybe(lp + "pre_norm.beta");
        L.g2 = get(lp + "ff_norm.gamma").p;
        L.be2 = maybe(lp + "ff_norm.beta");
        if (fp8) {
          L.qkv8 = pack_linear_fp8(lp + "self_attn.to_qkv.weight", "", false, st);
          // (also in 16 bits: where the fused to_qkv + attention kernel applies, that block runs on fp16 operands and
          // hands its output to the fp8 out-projection as e4m3 + scales)
          L.qkv = pack_linear(lp + "self_attn.to_qkv.weight", "", false, st);
          L.out8 = pack_linear_fp8(lp + "self_attn.to_out.weight", "", false, st);
          L.ff1_8 = pack_linear_fp8(lp + "ff.ff.0.proj.weight", lp + "ff.ff.0.proj.bias", true, st);
          L.ff2_8 = pack_linear_fp8(lp + "ff.ff.2.weight", lp + "ff.ff.2.bias", false, st);
          if (fold_ln8) {
            // ff_norm folded into FF-in as in the 16-bit modes: gamma is multiplied in BEFORE the block quantisation, the
            // row sums are those of the dequantised bytes (what the MFMAs multiply), beta goes into the bias
            const DevTensor& w = get(lp + "ff.ff.0.proj.weight");
            Packed8& p = L.ff1f8;
            p.N = L.ff1_8.N;
            p.K = L.ff1_8.K;
            p.w = (unsigned char*)dmalloc((size_t)p.N * p.K);
            p.s = (unsigned char*)dmalloc((size_t)p.N * (p.K / 32));
            launch_pack_weight_fp8(w.p, p.w, p.s, p.N, p.K, 1, st, L.g2);
            L.ff1_colsum8 = (float*)dmalloc(sizeof(float) * p.N);
            launch_fp8_row_sum(p.w, p.s, p.N, p.K, L.ff1_colsum8, st);
            p.bias = L.ff1_8.bias;
            if (L.be2) {
              float* tmp = (float*)dmalloc(sizeof(float) * p.N);
              launch_bias_plus_wbeta(w.p, L.be2, maybe(lp + "ff.ff.0.proj.bias"), p.N, p.K, tmp, st);
              p.bias = (float*)dmalloc(sizeof(float) * p.N);
              launch_pack_bias_swiglu(tmp, p.bias, p.N, st);
            }
          }
          continue;
        }
        L.qkv = pack_linear(lp + "self_attn.to_qkv.weight", "", false, st);
        L.out = pack_linear(lp + "self_attn.to_out.weight", "", false, st);
        L.ff1 = pack_linear(lp + "ff.ff.0.proj.weight", lp + "ff.ff.0.proj.bias", true, st);
        L.ff2 = pack_linear(lp + "ff.ff.2.weight", lp + "ff.ff.2.bias", false, st);
        if (fold_ln) {
          // LN(x) W^T = rstd (x (W diag(gamma))^T - mean colsum) + (b + W beta): gamma into the packed weight, the
          // row sums from the ROUNDED packed values, beta into the bias (dit_forward: FF-in consumes raw x planes)
          const DevTensor& w = get(lp + "ff.ff.0.proj.weight");
          Packed& p = L.ff1f;
          p.N = (int)w.shape[0];
          p.K = p.Cin = (int)(w.numel / w.shape[0]);
          p.taps = 1;
          p.ps = (long)p.N * p.K;
          p.w = (op16_t*)dmalloc(sizeof(op16_t) * p.ps * P);
          launch_pack_weight(w.p, nullptr, p.w, p.ps, PL, PACK_LINEAR_SWIGLU, p.N, p.K, p.K, p.N, 1, 1, st, L.g2);
          L.ff1_colsum = (float*)dmalloc(sizeof(float) * p.N);
          launch_packed_row_sum(p.w, p.ps, PL, p.N, p.K, L.ff1_colsum, st);
          const float* b1 = maybe(lp + "ff.ff.0.proj.bias");
          p.bias = L.ff1.bias;
          if (L.be2) {
            float* tmp = (float*)dmalloc(sizeof(float) * p.N);
            launch_bias_plus_wbeta(w.p, L.be2, b1, p.N, p.K, tmp, st);
            p.bias = (float*)dmalloc(sizeof(float) * p.N);
            launch_pack_bias_swiglu(tmp, p.bias, p.N, st);
          }
          p.bias_mod = p.N;
        }
      }
    }
    if (cfg.score_kind == DSN_SCORE_NCSNPP) finalize_ncsnpp(st);
    const int nb = cfg.vae_n_blocks;
    const int ch = cfg.vae_channels;
    std::vector<int> mult(nb + 1, 1);
    for (int i = 0; i < nb; ++i) mult[i + 1] = cfg.vae_c_mults[i];
    if (cfg.vae_has_decoder) {
      const std::string dp = "vae.decoder.";
      dec_in = pack_conv(dp + "layers.0.", st);
      dec_blocks.resize(nb);
      int li = 1;
      for (int i = nb; i >= 1; --i, ++li) {
        VaeBlock& b = dec_blocks[li - 1];
        const std::string bp = dp + "layers." + std::to_string(li) + ".";
        b.cin = mult[i] * ch;
        b.cout = mult[i - 1] * ch;
        b.stride = cfg.vae_strides[i - 1];
        b.act = make_act(bp + "layers.0.", b.cin, st);
        b.conv = pack_convT(bp + "layers.1.", b.stride, st);
        const int dils[3] = {1, 3, 9};
        for (int j = 0; j < 3; ++j)
          b.ru[j] = make_ru(bp + "layers." + std::to_string(2 + j) + ".", b.cout, dils[j], st);
      }
      dec_final_act = make_act(dp + "layers." + std::to_string(li) + ".", mult[0] * ch, st);
      // final conv (Cout = 1 per io channel; io_channels == 1 on this path): fold weight norm -> [k][C]
      {
        const std::string fp = dp + "layers." + std::to_string(li + 1) + ".";
        const bool wn = has(fp + "weight_v");
        const DevTensor& v = get(fp + (wn ? "weight_v" : "weight"));
        if (v.shape[0] != 1) fail(DSN_EINVAL, "decoder io_channels must be 1 (got %ld)", v.shape[0]);
        const int C = (int)v.shape[1], kw = (int)v.shape[2];
        dec_out_taps = kw;
        std::vector<float> hv(v.numel), hg(1, 1.f);
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpy(hv.data(), v.p, sizeof(float) * v.numel, hipMemcpyDeviceToHost));
        double nrm = 1.0, g = 1.0;
        if (wn) {
          HIPCHK(hipMemcpy(hg.data(), get(fp + "weight_g").p, sizeof(float), hipMemcpyDeviceToHost));
          double ss = 0;
          for (float x : hv) ss += (double)x * x;
          nrm = sqrt(ss);
          g = hg[0];
        }
        std::vector<float> w(kw * C);
        for (int c = 0; c < C; ++c)
          for (int k = 0; k < kw; ++k) w[k * C + c] = (float)(hv[c * kw + k] * ((float)g / (float)nrm));
        dec_out_w = (float*)dmalloc(sizeof(float) * w.size());
        HIPCHK(hipMemcpy(dec_out_w, w.data(), sizeof(float) * w.size(), hipMemcpyHostToDevice));
      }
    }
    if (cfg.vae_has_encoder) {
      const std::string ep = "vae.encoder.";
      {
        // first conv: Cin = 1, fold weight norm on the host (tiny) -> [Cout][k]
        const std::string fp = ep + "layers.0.";
        const bool wn = has(fp + "weight_v");
        const DevTensor& v = get(fp + (wn ? "weight_v" : "weight"));
        if (v.shape[1] != 1) fail(DSN_EINVAL, "encoder in_channels must be 1");
        const int Cout = (int)v.shape[0], kw = (int)v.shape[2];
        std::vector<float> hv(v.numel), hg(Cout, 1.f);
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpy(hv.data(), v.p, sizeof(float) * v.numel, hipMemcpyDeviceToHost));
        if (wn) HIPCHK(hipMemcpy(hg.data(), get(fp + "weight_g").p, sizeof(float) * Cout, hipMemcpyDeviceToHost));
        for (int co = 0; co < Cout; ++co) {
          float s = 1.f;
          if (wn) {
            float ss = 0;
            for (int k = 0; k < kw; ++k) ss += hv[co * kw + k] * hv[co * kw + k];
            s = hg[co] / sqrtf(ss);
          }
          for (int k = 0; k < kw; ++k) hv[co * kw + k] *= s;
        }
        enc_in_w = (float*)dmalloc(sizeof(float) * hv.size());
        HIPCHK(hipMemcpy(enc_in_w, hv.data(), sizeof(float) * hv.size(), hipMemcpyHostToDevice));
        enc_in_b = maybe(fp + "bias");
      }
      enc_blocks.resize(nb);
      int li = 1;
      for (int i = 0; i < nb; ++i, ++li) {
        VaeBlock& b = enc_blocks[i];
        const std::string bp = ep + "layers." + std::to_string(li) + ".";
        b.cin = mult[i] * ch;
        b.cout = mult[i + 1] * ch;
        b.stride = cfg.vae_strides[i];
        const int dils[3] = {1, 3, 9};
        for (int j = 0; j < 3; ++j) b.ru[j] = make_ru(bp + "layers." + std::to_string(j) + ".", b.cin, dils[j], st);
        b.act = make_act(bp + "layers.3.", b.cin, st);
        b.conv = pack_conv(bp + "layers.4.", st);
      }
      enc_final_act = make_act(ep + "layers." + std::to_string(li) + ".", mult[nb] * ch, st);
      enc_out = pack_conv(ep + "layers." + std::to_string(li + 1) + ".", st);
    }
    HIPCHK(hipStreamSynchronize(st));
    check_unused(strict);
    finalized = true;
  }

  // ---------------------------------------------------------------- GEMM helper
  // dense conv-like contraction over channels-last planes
  GemmDesc base_desc(const op16_t* A, long a_ps, const Packed& w, int Bn, int rows_per_b, int Lin) {
    GemmDesc d;
    memset(&d, 0, sizeof d);
    d.A = A;
    d.a_ps = a_ps;
    d.W = w.w;
    d.w_ps = w.ps;
    d.M = Bn * rows_per_b;
    d.N = w.N;
    d.Cin = w.Cin;
    d.taps = w.taps;
    d.rows_per_b = rows_per_b;
    d.Lin = Lin;
    d.in_stride = 1;
    d.tap_dil = 1;
    d.in_pad = 0;
    d.in_bstride = (long)Lin * w.Cin;
    d.in_row_elems = w.Cin;
    d.out_bstride = (long)rows_per_b * w.N;
    d.out_row_elems = w.N;
    d.out_off = 0;
    d.out_limit = d.out_bstride;
    d.resid_bstride = d.out_bstride;
    d.resid_row_elems = d.out_row_elems;
    d.resid_off = d.out_off;
    d.bias = w.bias;
    d.bias_mod = w.bias_mod;
    d.out_scale = 1.f;
    d.act_mod = 1;
    return d;
  }
  // development: "bm,bn,nst,bk" from the environment overrides the tile heuristic of one call site
  void dev_tile(GemmDesc& d, const char* var) const {
    const char* v = getenv(var);
    int bm, bn, nstg, bk;
    if (v && P == 1 && sscanf(v, "%d,%d,%d,%d", &bm, &bn, &nstg, &bk) == 4 && d.Cin % bk == 0) {
      d.cfg_bm = bm;
      d.cfg_bn = bn;
      d.cfg_nst = nstg;
      d.cfg_bk = bk;
    }
  }
  void set_act(GemmDesc& d, const ActP& a) {
    d.act = a.kind;
    d.act_a = a.a;
    d.act_b = a.ib;
    d.act_mod = a.mod;
  }
  // split-K factor for a GEMM whose output tile count cannot fill the chip (see dit_forward)
  int pick_ksplit(const GemmDesc& d) const {
    static const char* env = getenv("DSN_KSPLIT");
    const int nkt = d.taps * (d.Cin / 32);
    const int tiles = cdiv(d.M, 128) * cdiv(d.N, 128);
    int k = env ? atoi(env) : (400 + tiles / 2) / tiles;
    k = std::min(k, std::min(8, nkt / 8));
    return std::max(k, 1);
  }
  // DSN_AUDIT=1 (tests): before a GEMM-family launch, every pointer of its descriptor is checked on the HOST against
  // the device allocation that contains it (hipMemGetAddressRange), over the exact extent the kernels' guards allow --
  // the largest (item, row, column) offsets of A, W, the fp32 / plane / slab outputs, residual, bias vectors, GroupNorm
  // and LayerNorm partials -- plus the shape preconditions the epilogues rely on (GroupNorm partials: whole 64-row
  // slices of one item per wave tile, N % 4 == 0; halo conv: tiles inside one image).  An out-of-range descriptor is an
  // error naming the field instead of a GPU memory fault.
  void audit_span(const char* what, const void* p, long lo_bytes, long hi_bytes, const GemmDesc& d) const {
    if (!p || hi_bytes <= lo_bytes) return;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess)
      fail(DSN_EINVAL, "audit: %s = %p is not inside a device allocation (M=%d N=%d Cin=%d taps=%d)", what, p, d.M, d.N,
           d.Cin, d.taps);
    const char* b0 = (const char*)base;
    const char* q = (const char*)p;
    if (q + lo_bytes < b0 || q + hi_bytes > b0 + size)
      fail(DSN_EINVAL, "audit: %s spans [%ld, %ld) bytes from its pointer but the allocation holds [%ld, %ld) "
           "(M=%d N=%d Cin=%d taps=%d rows_per_b=%d)", what, lo_bytes, hi_bytes, (long)(b0 - q), (long)(b0 + size - q),
           d.M, d.N, d.Cin, d.taps, d.rows_per_b);
  }
  void audit_desc(const GemmDesc& d) const {
    if (d.a_scale) return;  // fp8 descriptors address bytes pairwise: audited by their launcher's shape checks
    const long nb = cdiv(d.M, d.rows_per_b), rows_last = d.M - (nb - 1) * (long)d.rows_per_b;
    const long rows_max = nb > 1 ? d.rows_per_b : rows_last;
    const long e16 = sizeof(op16_t), e32 = sizeof(float);
    if (d.M <= 0 || d.N <= 0 || d.rows_per_b <= 0) fail(DSN_EINVAL, "audit: empty GEMM");
    audit_span("A", d.A, 0, ((nb - 1) * d.in_bstride + (long)(d.Lin - 1) * d.in_row_elems + d.Cin + (P - 1) * d.a_ps) * e16, d);
    audit_span("W", d.W, 0, ((long)d.N * d.taps * d.Cin + (P - 1) * d.w_ps) * e16, d);
    const long n_out = d.swiglu ? d.N / 2 : d.N;
    const long rel_hi = std::min<long>(d.out_limit, (rows_max - 1) * (long)d.out_row_elems + d.out_off + n_out);
    const long out_hi = (nb - 1) * d.out_bstride + rel_hi;
    const long ks = std::max(d.ksplit, 1);
    if (d.stat_out) {  // residual-stream producer: plain row-major [M][N]
      audit_span("out_f32", d.out_f32, 0, (long)d.M * d.N * e32, d);
      audit_span("out_planes", d.out_planes, 0, ((long)d.M * d.N + (P - 1) * d.out_ps) * e16, d);
      audit_span("stat_out", d.stat_out, 0, (long)d.M * d.stat_np * 2 * e32, d);
    } else {
      audit_span("out_f32", d.out_f32, 0, (out_hi + (ks - 1) * d.slab_stride) * e32, d);
      audit_span("out_planes", d.out_planes, 0, (out_hi + (P - 1) * d.out_ps) * e16, d);
    }
    audit_span("resid", d.resid, 0, ((nb - 1) * d.resid_bstride + (rows_max - 1) * (long)d.resid_row_elems + d.resid_off + d.N) * e32, d);
    if (d.resid && d.resid_off < 0) fail(DSN_EINVAL, "audit: negative residual offset");
    audit_span("bias", d.bias, 0, (long)d.bias_mod * e32, d);
    audit_span("bbias", d.bbias, 0, ((nb - 1) * (long)d.bbias_stride + d.N) * e32, d);
    if (d.act == DSN_ACT_SNAKE) {
      audit_span("act_a", d.act_a, 0, (long)d.act_mod * e32, d);
      audit_span("act_b", d.act_b, 0, (long)d.act_mod * e32, d);
    }
    if (d.gn_stats) {
      if (d.rows_per_b % 64 != 0 || d.N % 4 != 0 || d.M % d.rows_per_b != 0 || d.ksplit > 1)
        fail(DSN_EINVAL, "audit: GroupNorm partials need whole 64-row slices (rows_per_b=%d M=%d N=%d ksplit=%d)",
             d.rows_per_b, d.M, d.N, d.ksplit);
      audit_span("gn_stats", d.gn_stats, 0, nb * (d.rows_per_b / 64) * (long)(d.N / 4) * 2 * e32, d);
      if (nb * (d.rows_per_b / 64) * (long)(d.N / 4) * 2 > ncs_slot_floats)
        fail(DSN_EINVAL, "audit: GroupNorm partials overflow their statistics slot");
      if (d.gn_stats2) {
        if (d.gn_qoff2 < 0 || d.gn_qoff2 + d.N / 4 > d.gn_nq2 ||
            nb * (d.rows_per_b / 64) * (long)d.gn_nq2 * 2 > ncs_slot_floats)
          fail(DSN_EINVAL, "audit: concat GroupNorm partials outside their slot (nq2=%d qoff2=%d N=%d)", d.gn_nq2,
               d.gn_qoff2, d.N);
        audit_span("gn_stats2", d.gn_stats2, 0, nb * (d.rows_per_b / 64) * (long)d.gn_nq2 * 2 * e32, d);
      }
    }
    if (d.ln_stats) {
      audit_span("ln_stats", d.ln_stats, 0, (long)d.M * d.ln_np * 2 * e32, d);
      audit_span("ln_colsum", d.ln_colsum, 0, (long)d.N * e32, d);
    }
    if (d.rope_cos) {
      audit_span("rope_cos", d.rope_cos, 0, (long)d.rope_S * 32 * e32, d);
      audit_span("rope_sin", d.rope_sin, 0, (long)d.rope_S * 32 * e32, d);
    }
    if (d.img_w > 0 && (d.rows_per_b != d.img_w * d.img_h || d.taps != 9))
      fail(DSN_EINVAL, "audit: 2-D conv descriptor with rows_per_b != H*W");
  }
  void run(const GemmDesc& d, hipStream_t st, int panel_bn = 0, bool skinny = false) {
    static const bool audit = getenv("DSN_AUDIT") != nullptr;
    if (audit) audit_desc(d);
    ProfRec pr;
    if (profiling) {
      HIPCHK(hipEventCreate(&pr.a));
      HIPCHK(hipEventCreate(&pr.b));
      pr.flops = 2.0 * (double)d.M * (double)d.N * ((double)d.taps * (double)d.Cin + (d.sc_A ? (double)d.sc_Cin : 0.0));
      pr.tag = cur_tag;
      HIPCHK(hipEventRecord(pr.a, st));
    }
    static const bool use_v1 = getenv("DSN_IGEMM_V1") != nullptr;
    // NCSN++ convs of single mixtures: a handful of 128 x 128 tiles each walking a long K alone -> split-K over enough
    // workgroups for a quarter of the chip, then the slab epilogue (B = 1, T = 16: score call 3.07 -> see DESIGN 5)
    static const bool no_csplit = getenv("DSN_NO_CONV_SPLIT") != nullptr;
    const long ctiles = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
    if (!no_csplit && P == 1 && cfg.score_kind == DSN_SCORE_NCSNPP && !skinny && panel_bn == 0 && d.ksplit <= 1 &&
        d.Cin % 32 == 0 && d.N % 64 == 0 && ctiles <= 32 && d.taps * d.Cin >= 512 && !d.swiglu && !d.rope_cos &&
        (d.out_f32 || d.out_planes) && d.out_off >= 0 && d.tap_dil >= 0 && d.in_stride == 1 &&
        (d.img_w > 0 || (d.taps == 1 && d.in_pad == 0))) {
      const int nkt = d.taps * d.Cin / 32;
      const int ks = (int)std::max(2L, std::min<long>(std::min<long>(8, nkt / 4), 64 / ctiles));
      const long span = (long)cdiv(d.M, d.rows_per_b) * d.out_bstride;  // floats one slab covers: the output view
      float* slabs = wsbuf<float>("conv_slabs", span * 8);
      GemmDesc g = d;
      g.ksplit = ks;
      g.slab_stride = span;
      g.out_f32 = slabs;
      g.out_planes = nullptr;
      g.gn_stats = nullptr;
      g.gn_stats2 = nullptr;
      g.cfg_bm = 128;
      g.cfg_bn = 128;
      g.cfg_nst = 3;
      g.cfg_bk = 32;
      hipError_t e1 = igemm2_launch(g, PL, st);
      if (e1 == hipSuccess) e1 = igemm_slab_epilogue_launch(d, PL, slabs, ks, span, st);
      if (profiling) {
        HIPCHK(hipEventRecord(pr.b, st));
        prof.push_back(pr);
      }
      if (e1 != hipSuccess) fail(DSN_EHIP, "split conv launch failed: %s", hipGetErrorString(e1));
      return;
    }
    hipError_t e = skinny ? igemm_skinny_launch(d, PL, st)
                          : (panel_bn > 0 ? igemm_panel_launch(d, PL, panel_bn, st)
                                          : ((use_v1 && d.ksplit <= 1) ? igemm_launch(d, PL, st) : igemm2_launch(d, PL, st)));
    if (profiling) {
      HIPCHK(hipEventRecord(pr.b, st));
      prof.push_back(pr);
    }
    if (e != hipSuccess) fail(DSN_EHIP, "igemm launch failed: %s (M=%d N=%d Cin=%d taps=%d)", hipGetErrorString(e),
                              d.M, d.N, d.Cin, d.taps);
  }

  // descriptor of an fp8 (MX) row-major GEMM: A8 [M][K] bytes + scales [M][K/32], weight w
  GemmDesc fp8_desc(const unsigned char* A8, const unsigned char* SA, const Packed8& w, int M) {
    Packed pk;
    pk.w = reinterpret_cast<op16_t*>(w.w);
    pk.ps = 0;
    pk.N = w.N;
    pk.Cin = w.K / 2;  // byte pairs
    pk.K = w.K / 2;
    pk.taps = 1;
    pk.bias = w.bias;
    pk.bias_mod = w.N;
    GemmDesc d = base_desc(reinterpret_cast<const op16_t*>(A8), 0, pk, 1, M, M);
    d.a_scale = SA;
    d.w_scale = w.s;
    d.mx_kblocks = w.K / 32;
    return d;
  }
  void run_fp8(const GemmDesc& d, hipStream_t st, int bn) {
    ProfRec pr;
    if (profiling) {
      HIPCHK(hipEventCreate(&pr.a));
      HIPCHK(hipEventCreate(&pr.b));
      pr.flops = 2.0 * (double)d.M * (double)d.N * 2.0 * (double)d.Cin;
      pr.tag = cur_tag;
      HIPCHK(hipEventRecord(pr.a, st));
    }
    hipError_t e = igemm_panel_fp8_launch(d, bn, st);
    if (profiling) {
      HIPCHK(hipEventRecord(pr.b, st));
      prof.push_back(pr);
    }
    if (e != hipSuccess) fail(DSN_EHIP, "fp8 igemm launch failed: %s (M=%d N=%d K=%d rows=%d bn=%d)", hipGetErrorString(e),
                              d.M, d.N, 2 * d.Cin, d.panel_rows, bn);
  }

  // fused ResidualUnit (ru_fused.hip) for the 128-channel layers; `in` and `out` planes must differ
  bool ru_fusable(const ResUnit& r) const {
    static const bool off = getenv("DSN_NO_FUSED_RU") != nullptr;
    return !off && r.conv7.taps == 7 && r.conv7.Cin == 128 && r.conv7.N == 128 && r.conv1.taps == 1 &&
           r.conv1.Cin == 128 && r.conv1.N == 128 && r.dil <= 9;
  }
  void run_ru(const ResUnit& r, const op16_t* in, long ps, float* xf, float* out_f32, op16_t* out, const ActP& next,
              int S, long L, hipStream_t st) {
    RuDesc d;
    memset(&d, 0, sizeof d);
    d.A = in;
    d.a_ps = ps;
    d.X = xf;
    d.W7 = r.conv7.w;
    d.w7_ps = r.conv7.ps;
    d.b7 = r.conv7.bias;
    d.W1 = r.conv1.w;
    d.w1_ps = r.conv1.ps;
    d.b1 = r.conv1.bias;
    d.out_f32 = out_f32;
    d.out_planes = out;
    d.out_ps = ps;
    d.act_mid = r.act2.kind;
    d.mid_a = r.act2.a;
    d.mid_b = r.act2.ib;
    d.act_out = next.kind;
    d.out_a = next.a;
    d.out_b = next.ib;
    d.S = S;
    d.L = (int)L;
    d.dil = r.dil;
    ProfRec pr;
    if (profiling) {
      HIPCHK(hipEventCreate(&pr.a));
      HIPCHK(hipEventCreate(&pr.b));
      pr.flops = 2.0 * (double)S * (double)L * 128.0 * 128.0 * 8.0;
      // algorithmic HBM bytes: planes in, fp32 residual in, fp32 out (when kept), planes out
      pr.hbm_bytes = (double)S * (double)L * 128.0 * (2.0 * P + 4.0 + (out_f32 ? 4.0 : 0.0) + 2.0 * P);
      pr.tag = "vae.residual_unit_fused";
      pr.hbm_bound = true;
      HIPCHK(hipEventRecord(pr.a, st));
    }
    hipError_t e = ru_fused_launch(d, PL, st);
    if (profiling) {
      HIPCHK(hipEventRecord(pr.b, st));
      prof.push_back(pr);
    }
    if (e != hipSuccess) fail(DSN_EHIP, "fused residual unit launch failed: %s (S=%d L=%ld dil=%d)",
                              hipGetErrorString(e), S, L, r.dil);
  }

  // ---------------------------------------------------------------- DiT score
  // xt [B,n,Dl,T], t [B], mix [B,1,Dl,T] -> score token-major [B*T][n*Dl] in ws "sc"
  // timestep token: t [rows] -> to_timestep_embed(FourierFeatures(t)) fp32 [rows][D]  (dit.py: timestep_features,
  // to_timestep_embed).  The sampler calls it once for all N x B step times (time_cache) instead of per score call.
  void dit_time_embed(const float* t, int rows, float* out, hipStream_t st) {
    const int D = cfg.dit_embed_dim;
    op16_t* TF = wsbuf<op16_t>("dit_TF", (long)rows * 256 * P);
    op16_t* TE = wsbuf<op16_t>("dit_TE", (long)rows * D * P);
    launch_timestep_features(t, tf_w, rows, 128, TF, (long)rows * 256, PL, st);
    GemmDesc d = base_desc(TF, (long)rows * 256, t1, rows, 1, 1);
    d.out_planes = TE;
    d.out_ps = (long)rows * D;
    d.act = DSN_ACT_SILU;
    run(d, st);
    GemmDesc e = base_desc(TE, (long)rows * D, t2, rows, 1, 1);
    e.out_f32 = out;
    run(e, st);
  }
  // Per-sampler-pass cache of network time inputs: tv [N][B] step times -> rows of `data` ([N*B][width])
  struct TimeCache {
    const float* t0 = nullptr;  // first element of the cached time vector
    long rows = 0;
    float* data = nullptr;
    int width = 0;
    const float* find(const float* t, int B) const {
      if (!data || t < t0 || t + B > t0 + rows) return nullptr;
      return data + (long)(t - t0) * width;
    }
  } time_cache;

  static bool use_panel_ok(int D) { return getenv("DSN_NO_PANEL") == nullptr && D % 64 == 0; }
  float* dit_forward(const float* xt, const float* t, const float* mix, int B, int T, hipStream_t st) {
    const int n = cfg.n_src, Dl = cfg.latent_dim, D = cfg.dit_embed_dim, H = cfg.dit_heads;
    const int io = n * Dl, din = io + Dl, S = T + 1;
    const long Mt = (long)B * T, M = (long)B * S;
    op16_t* Up = wsbuf<op16_t>("dit_Up", Mt * din * P);
    float* X = wsbuf<float>("dit_X", M * D);
    op16_t* Ap = wsbuf<op16_t>("dit_Ap", M * D * P);
    op16_t* QKVp = wsbuf<op16_t>("dit_QKVp", M * 3 * D * P);
    op16_t* FF = wsbuf<op16_t>("dit_FF", M * 4 * D * P);
    float* SC = wsbuf<float>("sc", Mt * io);
    // DSN_PREC_FP8: the layer GEMMs' A operands as e4m3 bytes + E8M0 block scales (LayerNorm output / attention
    // output share one buffer, the SwiGLU hidden state has its own)
    unsigned char* A8 = fp8 ? wsbuf<unsigned char>("dit_A8", M * D) : nullptr;
    unsigned char* SA8 = fp8 ? wsbuf<unsigned char>("dit_SA8", M * D / 32) : nullptr;
    unsigned char* H8 = fp8 ? wsbuf<unsigned char>("dit_H8", M * 4 * D) : nullptr;
    unsigned char* SH8 = fp8 ? wsbuf<unsigned char>("dit_SH8", M * 4 * D / 32) : nullptr;
    op16_t* lnout = fp8 ? reinterpret_cast<op16_t*>(A8) : Ap;
    // folded ff_norm (single-plane modes, panels of at most 80 rows fill whole rounds): to_out runs WITHOUT split-K in
    // 128-column tiles, adds the residual itself and writes x' (fp32), its raw operand plane and per-row statistics;
    // FF-in then applies the LayerNorm algebraically in its epilogue -- the LayerNorm launch between them is gone
    // single mixtures and pairs (up to 80 token rows = 5 sub-tiles; DSN_SKINNY_MAX moves the limit, at most 128): weight-streaming skinny kernels, split-K 8 for
    // the two N = D GEMMs so that every CU streams a share of their weights.  Measured (scripts/score_time.py, one
    // score call): M = 33: 1.52 ms vs 2.27 ms with the panel kernels; M = 17 (config C1): 1.37 vs 2.02 ms; M = 9:
    // 1.28 vs 2.01 ms; M = 61 (4 sub-tiles): 1.77 vs 2.10 ms; M = 66 (5): 1.72 vs 2.07; M = 99 (7): 2.03 vs 2.08; M = 126 (8): 2.22 vs 2.10.  (The window used to start at 33 rows: below it the launcher picked the 1- / 2-sub-tile
    // instantiations, which run 2.5x slower than the 3-sub-tile one on the same data -- see igemm_skinny_launch.)
    static const bool no_skinny = getenv("DSN_NO_SKINNY") != nullptr;
    const char* skm = getenv("DSN_SKINNY_MIN");  // read per call (tests)
    const char* skx = getenv("DSN_SKINNY_MAX");
    const int skinny_min = skm ? atoi(skm) : 1, skinny_max = skx ? std::min(atoi(skx), 128) : 80;
    const bool skinny = !no_skinny && P == 1 && !fp8 && M >= skinny_min && M <= skinny_max && D % 256 == 0;
    const char* sks = getenv("DSN_SKINNY_KS");  // development: split-K of the skinny N = D GEMMs (default 8)
    const int skinny_ks = sks ? std::max(1, std::min(atoi(sks), 8)) : 8;
    int fold_rows = 0;
    if ((fold_ln || fold_ln8) && use_panel_ok(D) && !skinny) {
      for (int rounds = 1; rounds <= 4 && !fold_rows; ++rounds) {
        const int np = 256 * rounds / std::max(1, cdiv(D, 128));
        if (np >= 1 && cdiv(M, np) <= 80) fold_rows = cdiv(M, np);
      }
    }
    op16_t* Xp = (fold_rows && !fp8) ? wsbuf<op16_t>("dit_Xp", M * D) : nullptr;
    // (fp8: raw x' as e4m3 + block scales; NOT the to_out input buffer -- other column tiles still read those rows)
    unsigned char* X8 = (fold_rows && fp8) ? wsbuf<unsigned char>("dit_X8", M * D) : nullptr;
    unsigned char* SX8 = (fold_rows && fp8) ? wsbuf<unsigned char>("dit_SX8", M * D / 32) : nullptr;
    float* ST = fold_rows ? wsbuf<float>("dit_ST", M * (D / 64) * 2) : nullptr;
    const int rot = 32;  // max(dim_heads/2, 32) with 64-wide heads
    const bool new_rope = !ws.count("rope_cos_" + std::to_string(S));
    float* rc = wsbuf<float>("rope_cos_" + std::to_string(S), (long)S * rot);
    float* rs = wsbuf<float>("rope_sin_" + std::to_string(S), (long)S * rot);
    if (new_rope) launch_rope_tables(rc, rs, S, rot, st);

    launch_pack_tokens(xt, io, mix, Dl, B, T, nullptr, Up, Mt * din, PL, st);
    {  // X[b, 1+t] = (U + U Wpre^T) Win^T, one folded matrix (fold_dit_io)
      Tag tg(this, "dit.project_in");
      GemmDesc d = base_desc(Up, Mt * din, pin, B, T, T);
      d.out_f32 = X;
      d.out_bstride = (long)S * D;
      d.out_off = D;
      d.out_limit = (long)S * D;
      run(d, st);
    }
    {  // timestep token -> X[b, 0]
      const float* te = time_cache.find(t, B);
      if (!te) {
        float* tmp = wsbuf<float>("dit_te1", (long)B * D);
        Tag tg(this, "dit.time_embed");
        dit_time_embed(t, B, tmp, st);
        te = tmp;
      }
      launch_copy_rows(te, X, B, D, (long)S * D, st);
    }
    // Residual-stream GEMMs (out-proj, FF-out) have N = D only: at M ~ 2k rows that is too few
    // 128x128 tiles to fill 256 CUs, so they run split-K into fp32 slabs and the slab reduction
    // (+ bias + residual) is fused into the LayerNorm that follows.
    static const bool no_panel = getenv("DSN_NO_PANEL") != nullptr;
    const bool use_panel = !no_panel && (D % 64 == 0);
    // Short row panels for the three narrower GEMMs in the single-plane modes, sized so that panels x column tiles
    // x split-K is one balanced round of 256 workgroups: at M = 2112 out-proj 16 x 8 x split-K 2 and FF-out
    // 16 x 4 x split-K 4 (132-row panels, 9 sub-tiles) = 256, QKV 21 x 12 (104-row panels, 7 sub-tiles) = 252
    // (sweep: profiles/r01_gemm_sweep_dit_panel132.log).
    static const bool no_short = getenv("DSN_NO_SHORT_PANEL") != nullptr;
    // (also for small batches: many short panels keep every CU streaming weights -- B = 8: 197 -> 145 ms per step)
    const bool short_panel = use_panel && !no_short && P == 1;  // split modes: measured, no gain
    // panel height for a GEMM with `wg_per_panel` = column tiles x split-K workgroups per row panel: as many
    // panels as fill whole rounds of 256 CUs
    auto panel_rows_for = [&](int wg_per_panel, int max_rows = 272) {
      for (int rounds = 1;; ++rounds) {  // whole rounds of 256 workgroups, panels at most max_rows (<= 272) rows tall
        const int np = std::max(1, 256 * rounds / std::max(1, wg_per_panel));
        const int rows = (cdiv(M, np) + 7) / 8 * 8;
        if (rows <= max_rows) return rows;
      }
    };
    static const char* qkv_panel_env = getenv("DSN_QKV_PANEL");
    const int qkv_panel = (use_panel && qkv_panel_env) ? atoi(qkv_panel_env) : (short_panel ? 256 : 0);
    // Fused to_qkv + attention (single-plane 16-bit modes, 64-wide heads, panels of whole items up to 144 rows -- 240 for
    // one long item, the 2-stage-ring variant): as many items per panel as keep panels x heads at a full round of the
    // chip; small batches (fewer than half a round of workgroups) and the skinny window keep the separate kernels.
    static const bool no_qa = getenv("DSN_NO_QKV_FUSE") != nullptr;
    int qa_ipp = 0;
    if (!no_qa && P == 1 && !skinny && D == H * 64 && S <= qkv_attention_max_rows()) {
      int ipp = std::min(B, std::max(1, 144 / S));  // several items share a panel only in the 144-row tile
      while (ipp > 1 && cdiv(B, ipp) * H < 256) --ipp;
      if (cdiv(B, ipp) * H >= 128) qa_ipp = ipp;
      const char* force = getenv("DSN_QA_IPP");  // tests: force the fused kernel with this many items per panel
      if (force && atoi(force) >= 1 && atoi(force) * S <= qkv_attention_max_rows()) qa_ipp = atoi(force);
    }
    op16_t* AOp = qa_ipp ? wsbuf<op16_t>("dit_AOp", M * D) : nullptr;
    float* slabs = nullptr;
    int pend_n = 0;
    const float* pend_bias = nullptr;
    const long slab_stride = M * D;
    for (int i = 0; i < cfg.dit_depth; ++i) {
      const DitLayer& L = layers[i];
      // algorithmic bytes of the fused reduce + LayerNorm: x in/out (when slabs are pending), slabs in, planes out
      auto ln_bytes = [&](int np) { return (double)M * D * (4.0 * (np ? 2 : 1) + 4.0 * np + 2.0 * P); };
      prof_launch("dit.residual_norm", ln_bytes(pend_n), st, [&] {
        // (fused to_qkv + attention reads 16-bit planes, in the fp8 mode too)
        launch_residual_norm(X, slabs, pend_n, slab_stride, pend_bias, L.g1, L.be1, qa_ipp ? Ap : lnout, M * D, PL,
                             (int)M, D, 1e-5f, 1, st, qa_ipp ? nullptr : SA8);
      });
      if (qa_ipp) {
        // to_qkv + rotary + attention in one launch (qkv_attn.hip): panels of qa_ipp whole items x one head
        QkvAttnDesc q;
        memset(&q, 0, sizeof q);
        q.A = Ap;
        q.W = L.qkv.w;
        q.bias = L.qkv.bias;
        q.rope_cos = rc;
        q.rope_sin = rs;
        q.out = fp8 ? nullptr : AOp;
        q.out8 = fp8 ? A8 : nullptr;       // fp8 mode: e4m3 + E8M0 scales for the fp8 out-projection
        q.out8_scale = fp8 ? SA8 : nullptr;
        q.M = (int)M;
        q.D = D;
        q.H = H;
        q.S = S;
        q.ipp = qa_ipp;
        q.q_scale = 0.125f;
        ProfRec pr;
        if (profiling) {
          HIPCHK(hipEventCreate(&pr.a));
          HIPCHK(hipEventCreate(&pr.b));
          pr.flops = 2.0 * (double)M * 3.0 * D * D + 4.0 * (double)B * H * (double)S * S * 64.0;
          pr.tag = "dit.qkv_attention";
          HIPCHK(hipEventRecord(pr.a, st));
        }
        const hipError_t e = qkv_attention_launch(q, PL, st);
        if (profiling) {
          HIPCHK(hipEventRecord(pr.b, st));
          prof.push_back(pr);
        }
        if (e != hipSuccess) fail(DSN_EHIP, "qkv_attention launch failed: %s", hipGetErrorString(e));
      } else {
      {  // q|k|v operand planes: rotary + 1/sqrt(dh) fused into the epilogue
        Tag tg(this, "dit.qkv");
        GemmDesc d = fp8 ? fp8_desc(A8, SA8, L.qkv8, (int)M) : base_desc(Ap, M * D, L.qkv, 1, (int)M, (int)M);
        d.out_planes = QKVp;
        d.out_ps = M * 3 * D;
        d.rope_cos = rc;
        d.rope_sin = rs;
        d.rope_S = S;
        d.qkv_D = D;
        d.q_scale = 0.125f;
        d.m_fast = 1;
        if (qkv_panel) {
          const int np = cdiv(M, 272);
          d.panel_rows = short_panel ? panel_rows_for(cdiv(3 * D, qkv_panel)) : (cdiv(M, np) + 7) / 8 * 8;
        }
        if (fp8) {
          d.panel_rows = panel_rows_for(cdiv(3 * D, 256), 208);
          run_fp8(d, st, 256);
        } else if (skinny) {
          run(d, st, 0, true);
        } else {
          run(d, st, qkv_panel);
        }
      }
      prof_launch("dit.attention", (double)M * D * 2.0 * P * 4.0, st,
                  [&] { launch_attention_mfma(QKVp, M * 3 * D, lnout, M * D, PL, B, S, H, 64, st, SA8); });
      }
      {
        Tag tg(this, "dit.attn_out");
        GemmDesc d = fp8 ? fp8_desc(A8, SA8, L.out8, (int)M)
                         : base_desc(qa_ipp ? AOp : Ap, M * D, L.out, 1, (int)M, (int)M);
        if (fold_rows) {
          d.panel_rows = fold_rows;
          d.resid = X;
          d.out_f32 = X;
          d.out_planes = Xp;
          d.out_fp8 = X8;
          d.out_fp8_scale = SX8;
          d.out_ps = M * D;
          d.stat_out = ST;
          d.stat_np = D / 64;
          static const bool out_mfast = getenv("DSN_OUT_MFAST") != nullptr;
          d.m_fast = out_mfast ? 1 : 0;  // an XCD's share walks ACROSS the 8 column tiles of a few row panels: the whole
                                         // 2 MB weight and 4 panels fit its L2 (m_fast = 1: every XCD re-fetches all of A)
          if (fp8) run_fp8(d, st, 128);
          else run(d, st, 128);
          pend_n = 0;
          pend_bias = nullptr;
        } else {
        static const char* ocfg = getenv("DSN_OUT_CFG");  // "bn,ksplit" (development)
        int obn = 128, oks = 2;
        if (ocfg) sscanf(ocfg, "%d,%d", &obn, &oks);
        d.ksplit = skinny ? skinny_ks : ((short_panel || fp8) ? oks : pick_ksplit(d));
        if (short_panel || fp8) d.panel_rows = panel_rows_for(cdiv(D, obn) * oks, (fp8 && obn == 256) ? 208 : 272);
        if (d.ksplit > 1) {
          slabs = wsbuf<float>("dit_slabs", slab_stride * 8);
          d.out_f32 = slabs;
          d.slab_stride = slab_stride;
          d.bias = nullptr;
        } else {
          d.resid = X;
          d.out_f32 = X;
        }
        if (fp8) run_fp8(d, st, obn);
        else if (skinny) run(d, st, 0, true);
        else run(d, st, short_panel ? obn : 0);
        pend_n = d.ksplit > 1 ? d.ksplit : 0;
        pend_bias = nullptr;
        }
      }
      if (!fold_rows)
        prof_launch("dit.residual_norm", ln_bytes(pend_n), st, [&] {
          launch_residual_norm(X, slabs, pend_n, slab_stride, pend_bias, L.g2, L.be2, lnout, M * D, PL, (int)M, D,
                               1e-5f, 1, st, SA8);
        });
      {
        // FF-in through the row-panel kernel: ceil(M/272) equal row panels x 256-column tiles -- for the
        // benchmark shape (M = 2112 -> 8 panels of 264 rows, N = 8192) exactly 256 workgroups, one round.
        Tag tg(this, "dit.ff_in");
        GemmDesc d = fp8 ? (fold_rows ? fp8_desc(X8, SX8, L.ff1f8, (int)M) : fp8_desc(A8, SA8, L.ff1_8, (int)M))
                         : (fold_rows ? base_desc(Xp, M * D, L.ff1f, 1, (int)M, (int)M)
                                      : base_desc(Ap, M * D, L.ff1, 1, (int)M, (int)M));
        d.swiglu = 1;
        if (fold_rows) {
          d.ln_stats = ST;
          d.ln_np = D / 64;
          d.ln_colsum = fp8 ? L.ff1_colsum8 : L.ff1_colsum;
          d.ln_eps = 1e-5f;
        }
        if (fp8) {
          d.out_fp8 = H8;
          d.out_fp8_scale = SH8;
        } else {
          d.out_planes = FF;
        }
        d.out_ps = M * 4 * D;
        d.out_bstride = M * 4L * D;
        d.out_row_elems = 4 * D;
        d.out_limit = d.out_bstride;
        d.m_fast = 1;
        if (use_panel) {
          const int np = cdiv(M, 272);
          d.panel_rows = short_panel ? panel_rows_for(cdiv(4 * D * 2, 256)) : (cdiv(M, np) + 7) / 8 * 8;
        }
        if (fp8) {
          // 256-column fp8 tiles up to 17 row sub-tiles (8 waves x 256 registers, branch-free main loop): one round at C2
          static const char* f8rows = getenv("DSN_FP8_FF1_ROWS");  // development: cap of the panel height (144: round 2)
          d.panel_rows = panel_rows_for(cdiv(4 * D * 2, 256), f8rows ? atoi(f8rows) : 272);
          run_fp8(d, st, 256);
        } else if (skinny) {
          run(d, st, 0, true);
        } else {
          run(d, st, use_panel ? 256 : 0);
        }
      }
      {
        Tag tg(this, "dit.ff_out");
        GemmDesc d = fp8 ? fp8_desc(H8, SH8, L.ff2_8, (int)M) : base_desc(FF, M * 4 * D, L.ff2, 1, (int)M, (int)M);
        static const char* fcfg = getenv("DSN_FF2_CFG");
        int fbn = 256, fks = 4;
        if (fcfg) sscanf(fcfg, "%d,%d", &fbn, &fks);
        d.ksplit = skinny ? skinny_ks : ((short_panel || fp8) ? fks : pick_ksplit(d));
        if (short_panel || fp8) d.panel_rows = panel_rows_for(cdiv(D, fbn) * fks, (fp8 && fbn == 256) ? 208 : 272);
        if (d.ksplit > 1) {
          slabs = wsbuf<float>("dit_slabs", slab_stride * 8);
          d.out_f32 = slabs;
          d.slab_stride = slab_stride;
          pend_bias = d.bias;
          d.bias = nullptr;
        } else {
          d.resid = X;
          d.out_f32 = X;
          pend_bias = nullptr;
        }
        if (fp8) run_fp8(d, st, fbn);
        else if (skinny) run(d, st, 0, true);
        else run(d, st, short_panel ? fbn : 0);
        pend_n = d.ksplit > 1 ? d.ksplit : 0;
      }
    }
    // final residual update + planes of X (no norm before project_out)
    prof_launch("dit.residual_norm", (double)M * D * (8.0 + 4.0 * pend_n + 2.0 * P), st, [&] {
      launch_residual_norm(X, slabs, pend_n, slab_stride, pend_bias, nullptr, nullptr, Ap, M * D, PL, (int)M, D, 1e-5f,
                           0, st);
    });
    {  // score = o + o Wpost^T with o = X[b, 1+t] Wout^T, one folded matrix
      Tag tg(this, "dit.project_out");
      GemmDesc d = base_desc(Ap, M * D, pout, B, T, S);
      d.in_pad = -1;
      d.in_bstride = (long)S * D;
      d.out_f32 = SC;
      // N = n_src * latent_dim is ONE 128-column tile: 128-row tiles leave B*T/128 (17 at C2) workgroups walking the
      // whole K alone; 64-row panels x 8 waves with a 4-stage ring double the workgroups and the loads in flight
      static const bool po_tile = getenv("DSN_POUT_TILE") != nullptr;
      if (use_panel && P == 1 && !po_tile && d.N <= 128 && d.M >= 1024) {
        d.panel_rows = 64;
        run(d, st, 128);
      } else {
        run(d, st);
      }
    }
    return SC;
  }

#include "engine_ncsnpp.inc"

  float* score_tokens(const float* xt, const float* t, const float* mix, int B, int T, hipStream_t st) {
    if (!finalized) fail(DSN_ESTATE, "weights not finalized");
    if (cfg.score_kind == DSN_SCORE_DIT) return dit_forward(xt, t, mix, B, T, st);
    if (cfg.score_kind == DSN_SCORE_NCSNPP) {
      Tag tg(this, "ncsnpp.conv");
      return ncsnpp_forward(xt, t, mix, B, T, st);
    }
    fail(DSN_ESTATE, "no score network configured (score_kind=%d)", cfg.score_kind);
  }

  // ---------------------------------------------------------------- OUVE scalars
  struct Sched {
    std::vector<float> t, std, step, gain, G, g;
    float stdT;
  };
  float ouve_std(float t) const {
    const double th = cfg.sde_theta, smin = cfg.sde_sigma_min, ls = log((double)cfg.sde_sigma_max / smin);
    const float a = expf((float)(-2.0 * th) * t);
    const float b = expf((float)(2.0 * (th + ls)) * t) - 1.f;
    const float num = (float)(smin * smin) * a * b * (float)ls;
    return sqrtf(num / (float)(th + ls));
  }
  std::vector<float> t_override;  // scheduled sampler: explicit timesteps (first N used), empty = linspace
  Sched schedule(int N, float t_eps, float snr) const {
    Sched s;
    const double smin = cfg.sde_sigma_min, smax = cfg.sde_sigma_max, ls = log(smax / smin);
    s.t.resize(N);
    s.std.resize(N);
    s.step.resize(N);
    s.gain.resize(N);
    s.G.resize(N);
    s.g.resize(N);
    // torch.linspace(1, eps, N) in fp32: step = (end-start)/(N-1); symmetric fill
    const float start = 1.f, end = t_eps;
    const float stp = N > 1 ? (end - start) / (float)(N - 1) : 0.f;
    const int halfway = N / 2;
    for (int i = 0; i < N; ++i) s.t[i] = i < halfway ? start + stp * (float)i : end - stp * (float)(N - 1 - i);
    if ((int)t_override.size() >= N)
      for (int i = 0; i < N; ++i) s.t[i] = t_override[i];
    const float sqdt = sqrtf((float)(1.0 / N));
    for (int i = 0; i < N; ++i) {
      const float t = s.t[i];
      s.std[i] = ouve_std(t);
      const float q = snr * s.std[i];
      s.step[i] = q * q * 2.f;
      s.gain[i] = sqrtf(s.step[i] * 2.f);
      const float sigma = (float)smin * powf((float)(smax / smin), t);
      s.g[i] = sigma * (float)sqrt(2.0 * ls);
      s.G[i] = s.g[i] * sqdt;
    }
    s.stdT = ouve_std(1.f);
    return s;
  }

  // ---------------------------------------------------------------- sampler
  // y [B,1,Dl,T]; noise [(1+N(c+1))][B,n,Dl,T]; returns device pointer of the result
  // vec_t = ones(B) * timesteps[i] for every step, uploaded once per (B, schedule)
  void upload_timesteps(int B, int N, float t_eps, float snr, hipStream_t st) {
    const Sched s = schedule(N, t_eps, snr);
    std::vector<float> ht((size_t)B * N);
    for (int i = 0; i < N; ++i)
      for (int b = 0; b < B; ++b) ht[(size_t)i * B + b] = s.t[i];
    float* tv = wsbuf<float>("pc_t", (long)B * N);
    if (ht == tv_host && tv_B == B) return;
    HIPCHK(hipMemcpyAsync(tv, ht.data(), sizeof(float) * ht.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    tv_host = ht;
    tv_B = B;
  }
  struct PcOpts {
    int pred = DSN_PRED_REVERSE_DIFFUSION, corr = DSN_CORR_ALD, c = 1, denoise = 1;
    float snr = 0.5f, t_eps = 0.03f;
    const float* prior_mean = nullptr;
    float* inter = nullptr;
    long draws(int N) const { return 1 + (long)N * (c + (pred == DSN_PRED_NONE ? 0 : 1)); }
  };
  float* pc_sample(const float* y, const float* noise, int B, int T, int N, const PcOpts& o, hipStream_t st) {
    const int n = cfg.n_src, Dl = cfg.latent_dim;
    const long sz = (long)B * n * Dl * T;
    float* x = wsbuf<float>("pc_x", sz);
    float* xm = wsbuf<float>("pc_xm", sz);
    float* tv = wsbuf<float>("pc_t", (long)B * N);
    float* norms = o.corr == DSN_CORR_LANGEVIN ? wsbuf<float>("pc_norms", 2L * B) : nullptr;
    const Sched s = schedule(N, o.t_eps, o.snr);
    const float dt = (float)(1.0 / N);
    const float* z = noise;
    struct CacheScope {  // the cache is only valid while this pass (or its graph capture) runs
      TimeCache& c;
      ~CacheScope() { c = TimeCache(); }
    } cache_scope{time_cache};
    {  // every step's time embedding in one batch, off the per-call path (-1.3 % sampler time)
      const bool dit = cfg.score_kind == DSN_SCORE_DIT;
      const int width = dit ? cfg.dit_embed_dim : ncs_dense_total;
      float* all = wsbuf<float>("te_all", (long)N * B * width);
      Tag tg(this, "score.time_embed");
      if (dit) dit_time_embed(tv, N * B, all, st);
      else ncs_time_dense(tv, N * B, all, st);
      time_cache.t0 = tv;
      time_cache.rows = (long)N * B;
      time_cache.data = all;
      time_cache.width = width;
    }
    launch_pc_prior(o.prior_mean ? o.prior_mean : y, o.prior_mean != nullptr, z, x, s.stdT, B, n, Dl, T, st);
    z += sz;
    const bool keep_mean = o.inter != nullptr;  // only `intermediate` reads the corrector's x_mean
    for (int i = 0; i < N; ++i) {
      const float* ti = tv + (long)i * B;
      for (int k = 0; k < o.c; ++k) {
        float* sc = score_tokens(x, ti, y, B, T, st);
        if (norms) {
          launch_pc_item_norms(sc, (long)n * Dl * T, B, norms, st);
          launch_pc_item_norms(z, (long)n * Dl * T, B, norms + B, st);
        }
        launch_pc_corrector(x, keep_mean ? xm : nullptr, sc, z, s.step[i], s.gain[i], norms, o.snr, B, n, Dl, T, st);
        z += sz;
      }
      if (o.inter) {  // (xt, xt_mean) as the corrector returned them; c == 0 / a fresh loop: x_mean = x
        if (o.c == 0) HIPCHK(hipMemcpyAsync(xm, x, sizeof(float) * sz, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(o.inter + (2L * i) * sz, x, sizeof(float) * sz, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(o.inter + (2L * i + 1) * sz, xm, sizeof(float) * sz, hipMemcpyDeviceToDevice, st));
      }
      if (o.pred == DSN_PRED_NONE) {  // x, x_mean = x, x
        if (o.denoise && i == N - 1) HIPCHK(hipMemcpyAsync(xm, x, sizeof(float) * sz, hipMemcpyDeviceToDevice, st));
        continue;
      }
      float* sc = score_tokens(x, ti, y, B, T, st);
      launch_pc_predictor(x, xm, y, sc, z, cfg.sde_theta, dt, s.G[i], s.g[i], o.pred == DSN_PRED_EULER_MARUYAMA, B, n,
                          Dl, T, st);
      z += sz;
    }
    return o.denoise ? xm : x;
  }

  // ---------------------------------------------------------------- secondary sampler family
  // MixSDE / PriorMixSDE predictor-corrector loop with the ald2 corrector (reference sdes.py:182-593,
  // correctors.py:87-121, __init__.py:133-193) on the latent state read as [B, n, D*T]; scalar schedules in fp32 as
  // torch evaluates them on a [B]-vector of equal times.
  struct MixOpts {
    int prior_mix = 0, avg_len = 510, pred = DSN_PRED_REVERSE_DIFFUSION, corr = DSN_MIXCORR_ALD2, c = 1, denoise = 1;
    float d_lambda = 2.f, sigma_min = 0.05f, sigma_max = 0.5f, snr = 0.5f, t_eps = 0.03f;
    long draws(int N) const { return 1 + (long)N * (c + (pred == DSN_PRED_NONE ? 0 : 1)); }
  };
  static void mix_eigval(const MixOpts& o, float t, float& ev1, float& ev2) {
    const double ratio = (double)o.sigma_max / o.sigma_min, logsig = log(ratio);
    const float mult = (float)((double)o.sigma_min * o.sigma_min);
    const float srp = powf((float)ratio, 2.f * t);
    ev1 = mult * (srp - 1.f);
    ev2 = mult * (srp - expf((float)(-2.0 * o.d_lambda) * t)) / (float)(1.0 + o.d_lambda / logsig);
  }
  float* pc_sample_mix(const float* y, const float* noise, int B, int T, int N, const MixOpts& o, hipStream_t st) {
    const int n = cfg.n_src, Dl = cfg.latent_dim;
    const long sz = (long)B * n * Dl * T;
    float* x = wsbuf<float>("pc_x", sz);
    float* xm = wsbuf<float>("pc_xm", sz);
    float* tv = wsbuf<float>("pc_t", (long)B * N);
    float* smix = o.prior_mix ? wsbuf<float>("pc_smix", (long)B * Dl * T) : nullptr;
    const Sched s = schedule(N, o.t_eps, o.snr);  // timesteps = linspace(1, eps, N)
    const float dt = (float)(1.0 / N), sqdt = sqrtf(dt);
    const double ratio = (double)o.sigma_max / o.sigma_min, logsig = log(ratio);
    struct CacheScope {
      TimeCache& c;
      ~CacheScope() { c = TimeCache(); }
    } cache_scope{time_cache};
    {
      const bool dit = cfg.score_kind == DSN_SCORE_DIT;
      const int width = dit ? cfg.dit_embed_dim : ncs_dense_total;
      float* all = wsbuf<float>("te_all", (long)N * B * width);
      Tag tg(this, "score.time_embed");
      if (dit) dit_time_embed(tv, N * B, all, st);
      else ncs_time_dense(tv, N * B, all, st);
      time_cache.t0 = tv;
      time_cache.rows = (long)N * B;
      time_cache.data = all;
      time_cache.width = width;
    }
    if (smix) launch_sigma_mix(y, smix, B, Dl * T, o.avg_len, st);
    const float* z = noise;
    float ev1, ev2;
    mix_eigval(o, 1.f, ev1, ev2);
    launch_mix_prior(y, z, x, smix, sqrtf(ev1), sqrtf(ev2), B, n, Dl, T, st);
    z += sz;
    HIPCHK(hipMemcpyAsync(xm, x, sizeof(float) * sz, hipMemcpyDeviceToDevice, st));
    for (int i = 0; i < N; ++i) {
      const float* ti = tv + (long)i * B;
      const float t = s.t[i];
      if (o.corr == DSN_MIXCORR_ALD2 && o.c > 0) {
        mix_eigval(o, t, ev1, ev2);
        for (int k = 0; k < o.c; ++k) {
          float* sc = score_tokens(x, ti, y, B, T, st);
          launch_mix_corrector(x, xm, sc, z, smix, sqrtf(ev1), sqrtf(ev2), o.snr, B, n, Dl, T, st);
          z += sz;
        }
      }
      if (o.pred == DSN_PRED_NONE) {
        HIPCHK(hipMemcpyAsync(xm, x, sizeof(float) * sz, hipMemcpyDeviceToDevice, st));
        continue;
      }
      const float g = (float)o.sigma_min * powf((float)ratio, t) * (float)sqrt(2.0 * logsig);
      float* sc = score_tokens(x, ti, y, B, T, st);
      launch_mix_predictor(x, xm, sc, z, smix, o.d_lambda, dt, g, sqdt, o.pred == DSN_PRED_EULER_MARUYAMA, B, n, Dl, T, st);
      z += sz;
    }
    return o.denoise ? xm : x;
  }

  // get_sb_sampler (reference src/sdes/__init__.py:284-389) with SBVESDE (sdes.py:701-779): xt = y repeated over
  // the sources, N first-order bridge steps over linspace(1, eps, N + 1); the score network's output is the data
  // estimate.  sde type consumes one noise tensor per step (the last step's weight is zero, the draw still happens).
  float* sb_sample(const float* y, const float* noise, int B, int T, int N, float k, float c, float sbeps, float t_eps,
                   int ode, hipStream_t st) {
    const int n = cfg.n_src, Dl = cfg.latent_dim;
    const long sz = (long)B * n * Dl * T;
    float* x = wsbuf<float>("pc_x", sz);
    float* tv = wsbuf<float>("pc_t", (long)B * N);
    struct CacheScope {
      TimeCache& c;
      ~CacheScope() { c = TimeCache(); }
    } cache_scope{time_cache};
    {
      const bool dit = cfg.score_kind == DSN_SCORE_DIT;
      const int width = dit ? cfg.dit_embed_dim : ncs_dense_total;
      float* all = wsbuf<float>("te_all", (long)N * B * width);
      Tag tg(this, "score.time_embed");
      if (dit) dit_time_embed(tv, N * B, all, st);
      else ncs_time_dense(tv, N * B, all, st);
      time_cache.t0 = tv;
      time_cache.rows = (long)N * B;
      time_cache.data = all;
      time_cache.width = width;
    }
    auto sig = [&](float t) { return sqrtf((c * (powf(k, 2.f * t) - 1.f)) / (2.f * logf(k))); };
    const std::vector<float> ts = sb_times(N, t_eps);
    const float sig_T = sig(1.f);
    float sig_p = sig(ts[0]), sigb_p = sqrtf(sig_T * sig_T - sig_p * sig_p + sbeps);
    launch_repeat_sources(y, x, B, n, Dl, T, st);
    const float* z = noise;
    for (int i = 0; i < N; ++i) {
      const float t = ts[i + 1];
      const float sig_t = sig(t), sigb_t = sqrtf(sig_T * sig_T - sig_t * sig_t + sbeps);
      float* est = score_tokens(x, tv + (long)i * B, y, B, T, st);
      if (!ode) {
        const float w_prev = sig_t * sig_t / (sig_p * sig_p + sbeps);
        const float tmp = 1.f - sig_t * sig_t / (sig_p * sig_p + sbeps);
        const float w_z = (i == N - 1) ? 0.f : sig_t * sqrtf(tmp);
        launch_sb_update(x, est, z, w_prev, tmp, w_z, 0, B, n, Dl, T, st);
        z += sz;
      } else {
        const float w_prev = sig_t * sigb_t / (sig_p * sigb_p + sbeps);
        const float w_est = 1.f / (sig_T * sig_T + sbeps) * (sigb_t * sigb_t - sigb_p * sig_t * sigb_t / (sig_p + sbeps));
        const float w_pm = 1.f / (sig_T * sig_T + sbeps) * (sig_t * sig_t - sig_p * sig_t * sigb_t / (sigb_p + sbeps));
        launch_sb_update(x, est, y, w_prev, w_est, w_pm, 1, B, n, Dl, T, st);
      }
      sig_p = sig_t;
      sigb_p = sigb_t;
    }
    return x;
  }
  // torch.linspace(1, eps, N + 1) in fp32 (symmetric fill), entries 0..N
  static std::vector<float> sb_times(int N, float t_eps) {
    std::vector<float> ts((size_t)N + 1);
    const int steps = N + 1;
    const float stp = (t_eps - 1.f) / (float)(steps - 1);
    for (int i = 0; i < steps; ++i) ts[i] = i < steps / 2 ? 1.f + stp * (float)i : t_eps - stp * (float)(steps - 1 - i);
    return ts;
  }

  // ---------------------------------------------------------------- decoder
  int hop() const {
    int h = 1;
    for (int i = 0; i < cfg.vae_n_blocks; ++i) h *= cfg.vae_strides[i];
    return h;
  }
  // est [S][Dl][T] (S = B*n) -> wav fp32 [S][hop*T] in ws "dec_wav"
  float* decode(const float* est, int S, int T, hipStream_t st) {
    if (!cfg.vae_has_decoder) fail(DSN_ESTATE, "decoder not configured");
    const int Dl = cfg.latent_dim;
    long maxel = (long)S * T * dec_in.N;
    {
      long L = T;
      for (auto& b : dec_blocks) {
        L *= b.stride;
        maxel = std::max(maxel, (long)S * L * b.cout);
      }
    }
    op16_t* zp = wsbuf<op16_t>("dec_z", (long)S * T * Dl * P);
    op16_t* pa = wsbuf<op16_t>("dec_pa", maxel * P);
    op16_t* pb = wsbuf<op16_t>("dec_pb", maxel * P);
    op16_t* ph = wsbuf<op16_t>("dec_ph", maxel * P);
    float* xf = wsbuf<float>("dec_x", maxel);
    launch_pack_tokens(est, Dl, nullptr, 0, S, T, nullptr, zp, (long)S * T * Dl, PL, st);
    long L = T;
    {
      Tag tg(this, "vae.dec_conv_in");
      GemmDesc d = base_desc(zp, (long)S * T * Dl, dec_in, S, T, T);
      d.in_pad = (dec_in.taps - 1) / 2;
      d.out_planes = pa;
      d.out_ps = (long)S * T * dec_in.N;
      set_act(d, dec_blocks[0].act);
      run(d, st);
    }
    long a_ps = (long)S * T * dec_in.N;
    for (size_t bi = 0; bi < dec_blocks.size(); ++bi) {
      const VaeBlock& b = dec_blocks[bi];
      const long Lo = L * b.stride;
      const long o_ps = (long)S * Lo * b.cout;
      {  // ConvTranspose1d as a 2-tap phase GEMM
        Tag tg(this, "vae.dec_convT");
        GemmDesc d = base_desc(pa, a_ps, b.conv, S, (int)L + 1, (int)L);
        d.tap_dil = -1;
        d.out_bstride = Lo * b.cout;
        d.out_row_elems = b.stride * b.cout;
        d.out_off = -((b.stride + 1) / 2) * b.cout;
        d.out_limit = Lo * b.cout;
        d.out_f32 = xf;
        d.out_planes = pb;
        d.out_ps = o_ps;
        set_act(d, b.ru[0].act0);
        {
          // Shallow-K phase GEMMs (the last up-sampling layers: K = 2 x 128 or 2 x 256): while the 256 x 256 kernel
          // spilled ~200 registers a 3-stage 256 x 128 tile was faster there (8.3 vs 9.2 ms over the 5 layers); with
          // the lean epilogue the default 256 x 256 x BK 64 tile wins again (6.2 vs 7.3 ms) -- override kept for sweeps
          if (d.taps * d.Cin <= 512) dev_tile(d, "DSN_CONVT_TILE");
          if (d.taps * d.Cin > 512) dev_tile(d, "DSN_CONVT_DEEP_TILE");
        }
        run(d, st);
      }
      for (int j = 0; j < 3; ++j) {
        const ResUnit& r = b.ru[j];
        const ActP& next = j < 2 ? b.ru[j + 1].act0 : (bi + 1 < dec_blocks.size() ? dec_blocks[bi + 1].act : dec_final_act);
        if (ru_fusable(r)) {  // pb -> ph, then the two trade places
          run_ru(r, pb, o_ps, xf, j < 2 ? xf : nullptr, ph, next, S, Lo, st);
          std::swap(pb, ph);
          continue;
        }
        Tag tg(this, "vae.residual_unit_2gemm");
        {
          GemmDesc d = base_desc(pb, o_ps, r.conv7, S, (int)Lo, (int)Lo);
          d.tap_dil = r.dil;
          d.in_pad = r.dil * (r.conv7.taps - 1) / 2;
          d.out_planes = ph;
          d.out_ps = o_ps;
          set_act(d, r.act2);
          dev_tile(d, "DSN_RU7_TILE");
          run(d, st);
        }
        {
          GemmDesc d = base_desc(ph, o_ps, r.conv1, S, (int)Lo, (int)Lo);
          d.resid = xf;
          d.out_planes = pb;
          d.out_ps = o_ps;
          if (j < 2) d.out_f32 = xf;
          set_act(d, next);
          dev_tile(d, "DSN_RU1_TILE");
          run(d, st);
        }
      }
      std::swap(pa, pb);
      a_ps = o_ps;
      L = Lo;
    }
    float* wav = wsbuf<float>("dec_wav", (long)S * L);
    prof_launch("vae.dec_conv_out", (double)S * L * (dec_blocks.back().cout * 2.0 * P + 4.0), st, [&] {
      launch_conv_out1(pa, a_ps, PL, dec_out_w, wav, S, (int)L, dec_blocks.back().cout, dec_out_taps,
                       cfg.vae_final_tanh, st);
    });
    return wav;
  }

  // ---------------------------------------------------------------- encoder
  // wav [S][L] (L multiple of hop) + noise [S][Dl][T] -> y [S][Dl][T]
  void encode(const float* wav, const float* noise, float* y, int S, int L, hipStream_t st) {
    float* enc = encode_body(wav, S, L, st);
    launch_vae_sample(enc, noise, y, S, cfg.latent_dim, L / hop(), st);
  }
  // wav [S][L] -> encoder output (mean ++ scale) channels-last [S][L/hop][2*Dl] in ws "enc_out"
  float* encode_body(const float* wav, int S, int L, hipStream_t st) {
    if (!cfg.vae_has_encoder) fail(DSN_ESTATE, "encoder not configured");
    const int Dl = cfg.latent_dim, c0 = cfg.vae_channels;
    long maxel = (long)S * L * c0;
    {
      long l = L;
      for (auto& b : enc_blocks) {
        l /= b.stride;
        maxel = std::max(maxel, (long)S * l * b.cout);
      }
    }
    op16_t* pa = wsbuf<op16_t>("enc_pa", maxel * P);
    op16_t* ph = wsbuf<op16_t>("enc_ph", maxel * P);
    op16_t* pb = wsbuf<op16_t>("enc_pb", maxel * P);
    float* xf = wsbuf<float>("enc_x", maxel);
    long l = L;
    long a_ps = (long)S * l * c0;
    {
      const ActP& a = enc_blocks[0].ru[0].act0;
      launch_conv_in1(wav, enc_in_w, enc_in_b, S, L, c0, 7, xf, pa, a_ps, PL, a.kind, a.a, a.ib, st);
    }
    for (size_t bi = 0; bi < enc_blocks.size(); ++bi) {
      const VaeBlock& b = enc_blocks[bi];
      for (int j = 0; j < 3; ++j) {
        const ResUnit& r = b.ru[j];
        const ActP& next = j < 2 ? b.ru[j + 1].act0 : b.act;
        if (ru_fusable(r)) {  // pa -> ph, then the two trade places
          run_ru(r, pa, a_ps, xf, j < 2 ? xf : nullptr, ph, next, S, l, st);
          std::swap(pa, ph);
          continue;
        }
        Tag tg(this, "vae.residual_unit_2gemm");
        {
          GemmDesc d = base_desc(pa, a_ps, r.conv7, S, (int)l, (int)l);
          d.tap_dil = r.dil;
          d.in_pad = r.dil * (r.conv7.taps - 1) / 2;
          d.out_planes = ph;
          d.out_ps = a_ps;
          set_act(d, r.act2);
          run(d, st);
        }
        {
          GemmDesc d = base_desc(ph, a_ps, r.conv1, S, (int)l, (int)l);
          d.resid = xf;
          d.out_planes = pa;
          d.out_ps = a_ps;
          if (j < 2) d.out_f32 = xf;
          set_act(d, next);
          run(d, st);
        }
      }
      const long lo = l / b.stride;
      const long o_ps = (long)S * lo * b.cout;
      {  // strided conv k = 2s
        Tag tg(this, "vae.enc_strided_conv");
        GemmDesc d = base_desc(pa, a_ps, b.conv, S, (int)lo, (int)l);
        d.in_stride = b.stride;
        d.in_pad = (b.stride + 1) / 2;
        d.out_planes = pb;
        d.out_ps = o_ps;
        if (bi + 1 < enc_blocks.size()) {
          d.out_f32 = xf;
          set_act(d, enc_blocks[bi + 1].ru[0].act0);
        } else {
          set_act(d, enc_final_act);
        }
        run(d, st);
      }
      std::swap(pa, pb);
      a_ps = o_ps;
      l = lo;
    }
    float* enc = wsbuf<float>("enc_out", (long)S * l * enc_out.N);
    {
      Tag tg(this, "vae.enc_conv_out");
      GemmDesc d = base_desc(pa, a_ps, enc_out, S, (int)l, (int)l);
      d.in_pad = (enc_out.taps - 1) / 2;
      d.out_f32 = enc;
      run(d, st);
    }
    if (enc_out.N != 2 * Dl) fail(DSN_EINVAL, "encoder latent %d != 2*latent_dim %d", enc_out.N, 2 * Dl);
    return enc;
  }

  // chunk / paste schedule of AudioAutoencoder.decode_audio / encode_audio (autoencoders.py:596-731), latent frames
  struct ChunkPaste {
    int src, t0, t1, c0, c1;
  };
  std::vector<ChunkPaste> chunk_plan(int total, int chunk, int overlap) {
    if (overlap < 0 || chunk <= overlap || total < chunk)
      fail(DSN_EINVAL, "chunked coding needs 0 <= overlap < chunk_size <= frames (got %d, %d, %d)", overlap, chunk, total);
    const int hopf = chunk - overlap, ol = overlap / 2;
    std::vector<int> starts;
    for (int a = 0; a + chunk <= total; a += hopf) starts.push_back(a);
    if (starts.back() + chunk != total) starts.push_back(total - chunk);
    std::vector<ChunkPaste> plan;
    for (size_t i = 0; i < starts.size(); ++i) {
      const bool last = i + 1 == starts.size();
      ChunkPaste c;
      c.src = starts[i];
      c.t1 = last ? total : (int)i * hopf + chunk;
      c.t0 = last ? total - chunk : (int)i * hopf;
      c.c0 = 0;
      c.c1 = chunk;
      if (i > 0) {
        c.t0 += ol;
        c.c0 += ol;
      }
      if (!last) {
        c.t1 -= ol;
        c.c1 -= ol;
      }
      plan.push_back(c);
    }
    return plan;
  }
  // est [S][Dl][T] -> wav [S][hop*T] in ws "dec_long": every chunk is an independent decode of `chunk` frames
  float* decode_chunked(const float* est, int S, int T, int chunk, int overlap, hipStream_t st) {
    const int Dl = cfg.latent_dim, h = hop();
    const auto plan = chunk_plan(T, chunk, overlap);
    float* zc = wsbuf<float>("dec_chunk_in", (long)S * Dl * chunk);
    float* out = wsbuf<float>("dec_long", (long)S * h * T);
    const long Lc = (long)h * chunk, Lt = (long)h * T;
    for (const auto& c : plan) {
      HIPCHK(hipMemcpy2DAsync(zc, sizeof(float) * chunk, est + c.src, sizeof(float) * T, sizeof(float) * chunk,
                              (size_t)S * Dl, hipMemcpyDeviceToDevice, st));
      char key[64];
      snprintf(key, sizeof key, "dec:%d:%d", S, chunk);
      float* w = nullptr;
      run_graphed(key, st, [&](hipStream_t s2) { w = decode(zc, S, chunk, s2); });
      if (!w) w = wsbuf<float>("dec_wav", (long)S * Lc);
      HIPCHK(hipMemcpy2DAsync(out + (long)c.t0 * h, sizeof(float) * Lt, w + (long)c.c0 * h, sizeof(float) * Lc,
                              sizeof(float) * (size_t)(c.t1 - c.t0) * h, S, hipMemcpyDeviceToDevice, st));
    }
    return out;
  }
  // wav [S][hop*T] -> stitched encoder output [S][T][2*Dl] in ws "enc_long"
  float* encode_chunked(const float* wav, int S, int T, int chunk, int overlap, hipStream_t st) {
    const int h = hop(), E = 2 * cfg.latent_dim;
    const auto plan = chunk_plan(T, chunk, overlap);
    float* wc = wsbuf<float>("enc_chunk_in", (long)S * h * chunk);
    float* out = wsbuf<float>("enc_long", (long)S * T * E);
    const long Lc = (long)h * chunk, Lt = (long)h * T;
    for (const auto& c : plan) {
      HIPCHK(hipMemcpy2DAsync(wc, sizeof(float) * Lc, wav + (long)c.src * h, sizeof(float) * Lt, sizeof(float) * Lc, S,
                              hipMemcpyDeviceToDevice, st));
      float* e = encode_body(wc, S, (int)Lc, st);
      HIPCHK(hipMemcpy2DAsync(out + (long)c.t0 * E, sizeof(float) * (size_t)T * E, e + (long)c.c0 * E,
                              sizeof(float) * (size_t)chunk * E, sizeof(float) * (size_t)(c.t1 - c.t0) * E, S,
                              hipMemcpyDeviceToDevice, st));
    }
    return out;
  }
};

// =========================================================================== C-ABI
namespace {
thread_local std::string g_create_err;

template <class F>
int guarded(dsn_ctx* ctx, F&& f) {
  if (!ctx) return DSN_EINVAL;
  try {
    HIPCHK(hipSetDevice(ctx->cfg.device));
    if (ctx->fin_err_host && *ctx->fin_err_host) {
      *ctx->fin_err_host = 0;
      fail(DSN_EHIP, "a GroupNorm hand-off wait of an earlier call gave up (the workgroups of one conv were not resident "
                     "together): that call's results are invalid; DSN_NO_GN_FIN=1 selects the separate GroupNorm pass");
    }
    f();
    return DSN_OK;
  } catch (const Err& e) {
    ctx->err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    ctx->err = e.what();
    return DSN_EINVAL;
  }
}
}  // namespace

extern "C" {

dsn_ctx* dsn_create(const dsn_config* cfg) {
  try {
    if (!cfg) fail(DSN_EINVAL, "null config");
    if (cfg->precision < DSN_PREC_BF16 || cfg->precision > DSN_PREC_FP8)
      fail(DSN_EINVAL, "precision must be one of DSN_PREC_{BF16,BF16X3,FP16,FP16X3,FP8}");
    if (cfg->vae_n_blocks < 0 || cfg->vae_n_blocks > DSN_MAX_VAE_BLOCKS) fail(DSN_EINVAL, "bad vae_n_blocks");
    if (cfg->n_src < 1 || cfg->latent_dim % 32 != 0) fail(DSN_EINVAL, "bad n_src / latent_dim");
    if (cfg->score_kind == DSN_SCORE_DIT &&
        (cfg->dit_heads <= 0 || cfg->dit_embed_dim <= 0 || cfg->dit_embed_dim % cfg->dit_heads != 0 ||
         cfg->dit_embed_dim / cfg->dit_heads != 64))
      fail(DSN_EINVAL, "DiT: only 64-wide attention heads are implemented (embed_dim %d / %d heads = %d)",
           cfg->dit_embed_dim, cfg->dit_heads, cfg->dit_heads > 0 ? cfg->dit_embed_dim / cfg->dit_heads : 0);
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) fail(DSN_EINVAL, "device %d out of range (%d GPUs)", cfg->device, ndev);
    HIPCHK(hipSetDevice(cfg->device));
    dsn_ctx* c = new dsn_ctx();
    c->cfg = *cfg;
    c->P = (cfg->precision == DSN_PREC_BF16X3 || cfg->precision == DSN_PREC_FP16X3) ? 2 : 1;
    c->PL = DSN_PL(c->P, cfg->precision >= DSN_PREC_FP16 ? 1 : 0);
    c->fp8 = cfg->precision == DSN_PREC_FP8;
    if (c->fp8 && cfg->score_kind == DSN_SCORE_DIT && cfg->dit_embed_dim % 128 != 0)
      fail(DSN_EINVAL, "DSN_PREC_FP8: embed_dim must be a multiple of 128 (got %d)", cfg->dit_embed_dim);
    return c;
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return nullptr;
  }
}

void dsn_destroy(dsn_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->cfg.device);
  (void)hipDeviceSynchronize();
  for (auto& g : ctx->graphs)
    if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
  for (int k = 0; k < 2; ++k) {
    if (ctx->own[k]) (void)hipStreamDestroy(ctx->own[k]);
    if (ctx->ev_in[k]) (void)hipEventDestroy(ctx->ev_in[k]);
    if (ctx->ev_out[k]) (void)hipEventDestroy(ctx->ev_out[k]);
  }
  for (auto& kv : ctx->raw) (void)hipFree(kv.second.p);
  for (void* p : ctx->allocs) (void)hipFree(p);
  for (auto& kv : ctx->ws) (void)hipFree(kv.second.first);
  for (auto& kv : ctx->gnf_sync) (void)hipFree(kv.second.first);
  if (ctx->fin_err_host) (void)hipHostFree(ctx->fin_err_host);
  delete ctx;
}

const char* dsn_last_error(const dsn_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int dsn_load_tensor(dsn_ctx* ctx, const char* name, const float* data, const int64_t* shape, int ndim,
                    int is_device) {
  return guarded(ctx, [&] {
    if (!name || !data || ndim < 0 || ndim > 8) fail(DSN_EINVAL, "dsn_load_tensor: bad arguments");
    DevTensor t;
    t.numel = 1;
    for (int i = 0; i < ndim; ++i) {
      t.shape.push_back(shape[i]);
      t.numel *= shape[i];
    }
    if (t.numel <= 0) fail(DSN_EINVAL, "%s: empty tensor", name);
    auto it = ctx->raw.find(name);
    if (it != ctx->raw.end()) {
      HIPCHK(hipFree(it->second.p));
      ctx->raw.erase(it);
    }
    HIPCHK(hipMalloc((void**)&t.p, sizeof(float) * t.numel));
    HIPCHK(hipMemcpy(t.p, data, sizeof(float) * t.numel, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    ctx->raw[name] = t;
    ctx->finalized = false;
  });
}

int dsn_finalize_weights(dsn_ctx* ctx) { return dsn_finalize_weights_ex(ctx, 1); }

int dsn_finalize_weights_ex(dsn_ctx* ctx, int strict) {
  return guarded(ctx, [&] { ctx->finalize(nullptr, strict != 0); });
}

int dsn_score(dsn_ctx* ctx, const float* xt, const float* t, const float* mix, float* out, int B, int T,
              void* stream) {
  return guarded(ctx, [&] {
    if (!xt || !t || !mix || !out || B <= 0 || T <= 0) fail(DSN_EINVAL, "dsn_score: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    float* sc = ctx->score_tokens(xt, t, mix, B, T, st);
    launch_unpack_tokens(sc, out, B, ctx->cfg.n_src * ctx->cfg.latent_dim, T, st);
    HIPCHK(hipGetLastError());
  });
}

int dsn_ouve_schedule(const dsn_ctx* ctx, int N, float t_eps, float snr, float* timesteps, float* std_,
                      float* corr_step, float* corr_gain, float* G, float* std_T) {
  if (!ctx || N <= 0) return DSN_EINVAL;
  const dsn_ctx::Sched s = ctx->schedule(N, t_eps, snr);
  for (int i = 0; i < N; ++i) {
    if (timesteps) timesteps[i] = s.t[i];
    if (std_) std_[i] = s.std[i];
    if (corr_step) corr_step[i] = s.step[i];
    if (corr_gain) corr_gain[i] = s.gain[i];
    if (G) G[i] = s.G[i];
  }
  if (std_T) *std_T = s.stdT;
  return DSN_OK;
}

int dsn_pc_sample_ex(dsn_ctx* ctx, const float* y, const float* noise, uint64_t seed, float* x_out, int B, int T,
                     int N, const dsn_sampler_opts* opts, int* nfe_out, void* stream) {
  return guarded(ctx, [&] {
    if (!y || !x_out || !opts || B <= 0 || T <= 0 || N <= 0 || opts->corrector_steps < 0)
      fail(DSN_EINVAL, "dsn_pc_sample: bad arguments");
    if (opts->predictor < 0 || opts->predictor > DSN_PRED_NONE || opts->corrector < 0 ||
        opts->corrector > DSN_CORR_LANGEVIN)
      fail(DSN_EINVAL, "dsn_pc_sample: unknown predictor %d / corrector %d", opts->predictor, opts->corrector);
    dsn_ctx::PcOpts o;
    o.pred = opts->predictor;
    o.corr = opts->corrector;
    o.c = opts->corrector_steps;
    o.denoise = opts->denoise;
    o.snr = opts->snr;
    o.t_eps = opts->timesteps ? opts->timesteps[N - 1] : opts->t_eps;
    o.inter = opts->intermediates;
    hipStream_t caller = (hipStream_t)stream;
    const int n = ctx->cfg.n_src, Dl = ctx->cfg.latent_dim;
    const long ysz = (long)B * Dl * T, sz = ysz * n;
    const long draws = o.draws(N);
    // free-form schedules / intermediates are not worth a graph cache entry each
    const bool graphs = ctx->use_graphs;
    if (opts->timesteps) ctx->t_override.assign(opts->timesteps, opts->timesteps + N);
    if (opts->timesteps || opts->intermediates) ctx->use_graphs = false;
    struct Restore {
      dsn_ctx* c;
      bool g;
      ~Restore() {
        c->use_graphs = g;
        c->t_override.clear();
      }
    } restore{ctx, graphs};
    // stable workspace copies of the caller's tensors (graph replay needs fixed pointers)
    float* yb = ctx->wsbuf<float>("pc_y", ysz);
    float* nz = ctx->wsbuf<float>("pc_noise", sz * draws);
    float* pm = opts->prior_mean ? ctx->wsbuf<float>("pc_prior_mean", sz) : nullptr;
    o.prior_mean = pm;
    ctx->upload_timesteps(B, N, o.t_eps, o.snr, caller);
    hipStream_t st = ctx->enter(caller);
    HIPCHK(hipMemcpyAsync(yb, y, sizeof(float) * ysz, hipMemcpyDeviceToDevice, st));
    if (pm) HIPCHK(hipMemcpyAsync(pm, opts->prior_mean, sizeof(float) * sz, hipMemcpyDeviceToDevice, st));
    if (noise)
      HIPCHK(hipMemcpyAsync(nz, noise, sizeof(float) * sz * draws, hipMemcpyDeviceToDevice, st));
    else
      launch_randn(nz, sz * draws, seed, 0, st);
    char key[192];
    snprintf(key, sizeof key, "pc:%d:%d:%d:%d:%a:%a:%d:%d:%d:%d", B, T, N, o.c, o.snr, o.t_eps, o.denoise, o.pred,
             o.corr, pm != nullptr);
    float* res = nullptr;
    ctx->run_graphed(key, st, [&](hipStream_t s2) { res = ctx->pc_sample(yb, nz, B, T, N, o, s2); });
    if (!res) res = ctx->wsbuf<float>(o.denoise ? "pc_xm" : "pc_x", sz);  // replayed graph: same buffers
    HIPCHK(hipMemcpyAsync(x_out, res, sizeof(float) * sz, hipMemcpyDeviceToDevice, st));
    ctx->leave(caller, st);
    if (nfe_out) *nfe_out = N * (o.c + 1);
    HIPCHK(hipGetLastError());
  });
}

int dsn_pc_sample(dsn_ctx* ctx, const float* y, const float* noise, uint64_t seed, float* x_out, int B, int T, int N,
                  int corrector_steps, float snr, float t_eps, int denoise, int* nfe_out, void* stream) {
  dsn_sampler_opts o;
  memset(&o, 0, sizeof o);
  o.corrector_steps = corrector_steps;
  o.snr = snr;
  o.t_eps = t_eps;
  o.denoise = denoise;
  return dsn_pc_sample_ex(ctx, y, noise, seed, x_out, B, T, N, &o, nfe_out, stream);
}

// get_pc_scheduled_sampler (src/sdes/__init__.py:49-130): same loop with caller-provided timesteps
// (linear / log / revlog grids of N+1 points, the first N are used).  The reference's `dt` stays 1/N in
// this sampler too (its `getattr(kwargs, "dt", ...)` on a dict never finds the key).
int dsn_pc_sample_sched(dsn_ctx* ctx, const float* y, const float* noise, uint64_t seed, float* x_out, int B, int T,
                        int N, const float* timesteps_host, int corrector_steps, float snr, int denoise, int* nfe_out,
                        void* stream) {
  if (!ctx || !timesteps_host || N <= 0) return DSN_EINVAL;
  dsn_sampler_opts o;
  memset(&o, 0, sizeof o);
  o.corrector_steps = corrector_steps;
  o.snr = snr;
  o.denoise = denoise;
  o.timesteps = timesteps_host;
  return dsn_pc_sample_ex(ctx, y, noise, seed, x_out, B, T, N, &o, nfe_out, stream);
}

int dsn_pc_sample_mix(dsn_ctx* ctx, const float* y, const float* noise, uint64_t seed, float* x_out, int B, int T, int N,
                      const dsn_mix_opts* opts, int* nfe_out, void* stream) {
  return guarded(ctx, [&] {
    if (!y || !x_out || !opts || B <= 0 || T <= 0 || N <= 0 || opts->corrector_steps < 0)
      fail(DSN_EINVAL, "dsn_pc_sample_mix: bad arguments");
    if (opts->predictor < 0 || opts->predictor > DSN_PRED_NONE || opts->corrector < 0 || opts->corrector > DSN_MIXCORR_NONE)
      fail(DSN_EINVAL, "dsn_pc_sample_mix: unknown predictor %d / corrector %d", opts->predictor, opts->corrector);
    const int n = ctx->cfg.n_src, Dl = ctx->cfg.latent_dim;
    if (n > 4) fail(DSN_EINVAL, "dsn_pc_sample_mix: at most 4 sources");
    if (!opts->prior_mix && n != 2)
      fail(DSN_EINVAL, "MixSDE.prior_sampling is written for 2 sources (reference sdes.py:347); use PriorMixSDE");
    if (opts->prior_mix && opts->avg_len < 1) fail(DSN_EINVAL, "PriorMixSDE: avg_len must be >= 1");
    if (!(opts->sigma_max > opts->sigma_min) || !(opts->sigma_min > 0)) fail(DSN_EINVAL, "need 0 < sigma_min < sigma_max");
    dsn_ctx::MixOpts o;
    o.prior_mix = opts->prior_mix;
    o.avg_len = opts->avg_len;
    o.pred = opts->predictor;
    o.corr = opts->corrector;
    o.c = opts->corrector == DSN_MIXCORR_NONE ? 0 : opts->corrector_steps;
    o.denoise = opts->denoise;
    o.d_lambda = opts->d_lambda;
    o.sigma_min = opts->sigma_min;
    o.sigma_max = opts->sigma_max;
    o.snr = opts->snr;
    o.t_eps = opts->t_eps;
    hipStream_t caller = (hipStream_t)stream;
    const long ysz = (long)B * Dl * T, sz = ysz * n, draws = o.draws(N);
    float* yb = ctx->wsbuf<float>("pc_y", ysz);
    float* nz = ctx->wsbuf<float>("pc_noise", sz * draws);
    ctx->upload_timesteps(B, N, o.t_eps, o.snr, caller);
    hipStream_t st = ctx->enter(caller);
    HIPCHK(hipMemcpyAsync(yb, y, sizeof(float) * ysz, hipMemcpyDeviceToDevice, st));
    if (noise) HIPCHK(hipMemcpyAsync(nz, noise, sizeof(float) * sz * draws, hipMemcpyDeviceToDevice, st));
    else launch_randn(nz, sz * draws, seed, 0, st);
    char key[224];
    snprintf(key, sizeof key, "pcmix:%d:%d:%d:%d:%d:%d:%d:%d:%a:%a:%a:%a:%a:%d", B, T, N, o.prior_mix, o.avg_len, o.pred,
             o.corr, o.c, o.d_lambda, o.sigma_min, o.sigma_max, o.snr, o.t_eps, o.denoise);
    float* res = nullptr;
    ctx->run_graphed(key, st, [&](hipStream_t s2) { res = ctx->pc_sample_mix(yb, nz, B, T, N, o, s2); });
    if (!res) res = ctx->wsbuf<float>(o.denoise ? "pc_xm" : "pc_x", sz);
    HIPCHK(hipMemcpyAsync(x_out, res, sizeof(float) * sz, hipMemcpyDeviceToDevice, st));
    ctx->leave(caller, st);
    if (nfe_out) *nfe_out = N * (o.c + 1);
    HIPCHK(hipGetLastError());
  });
}

int dsn_sb_sample(dsn_ctx* ctx, const float* y, const float* noise, uint64_t seed, float* x_out, int B, int T, int N, float k,
                  float c, float sb_eps, float t_eps, int sampler_type, void* stream) {
  return guarded(ctx, [&] {
    if (!y || !x_out || B <= 0 || T <= 0 || N <= 0) fail(DSN_EINVAL, "dsn_sb_sample: bad arguments");
    if (sampler_type != DSN_SB_SDE && sampler_type != DSN_SB_ODE) fail(DSN_EINVAL, "Invalid type. Choose 'ode' or 'sde'.");
    if (!(k > 0) || k == 1.f || !(c > 0)) fail(DSN_EINVAL, "SBVESDE: need k > 0, k != 1, c > 0");
    const int n = ctx->cfg.n_src, Dl = ctx->cfg.latent_dim;
    hipStream_t caller = (hipStream_t)stream;
    const long ysz = (long)B * Dl * T, sz = ysz * n;
    const long draws = sampler_type == DSN_SB_SDE ? N : 0;
    float* yb = ctx->wsbuf<float>("pc_y", ysz);
    float* nz = draws ? ctx->wsbuf<float>("pc_noise", sz * draws) : nullptr;
    {  // step times t_1 .. t_N of linspace(1, eps, N + 1) as the network's time input
      const std::vector<float> ts = dsn_ctx::sb_times(N, t_eps);
      std::vector<float> ht((size_t)B * N);
      for (int i = 0; i < N; ++i)
        for (int b = 0; b < B; ++b) ht[(size_t)i * B + b] = ts[i + 1];
      float* tv = ctx->wsbuf<float>("pc_t", (long)B * N);
      if (!(ht == ctx->tv_host && ctx->tv_B == B)) {
        HIPCHK(hipMemcpyAsync(tv, ht.data(), sizeof(float) * ht.size(), hipMemcpyHostToDevice, caller));
        HIPCHK(hipStreamSynchronize(caller));
        ctx->tv_host = ht;
        ctx->tv_B = B;
      }
    }
    hipStream_t st = ctx->enter(caller);
    HIPCHK(hipMemcpyAsync(yb, y, sizeof(float) * ysz, hipMemcpyDeviceToDevice, st));
    if (draws) {
      if (noise) HIPCHK(hipMemcpyAsync(nz, noise, sizeof(float) * sz * draws, hipMemcpyDeviceToDevice, st));
      else launch_randn(nz, sz * draws, seed, 0, st);
    }
    char key[160];
    snprintf(key, sizeof key, "sb:%d:%d:%d:%a:%a:%a:%a:%d", B, T, N, k, c, sb_eps, t_eps, sampler_type);
    float* res = nullptr;
    ctx->run_graphed(key, st, [&](hipStream_t s2) {
      res = ctx->sb_sample(yb, nz, B, T, N, k, c, sb_eps, t_eps, sampler_type == DSN_SB_ODE, s2);
    });
    if (!res) res = ctx->wsbuf<float>("pc_x", sz);
    HIPCHK(hipMemcpyAsync(x_out, res, sizeof(float) * sz, hipMemcpyDeviceToDevice, st));
    ctx->leave(caller, st);
    HIPCHK(hipGetLastError());
  });
}

int dsn_hop_length(const dsn_ctx* ctx) { return ctx ? ctx->hop() : DSN_EINVAL; }

int dsn_latent_frames(const dsn_ctx* ctx, int L) {
  if (!ctx || L < 0) return DSN_EINVAL;
  const int h = ctx->hop();
  return (L + (h - L % h)) / h;  // reference utils.pad: a full extra hop when L % hop == 0
}

int dsn_decode(dsn_ctx* ctx, const float* est, float* wav, int B, int T, int target_len, void* stream) {
  return guarded(ctx, [&] {
    if (!est || !wav || B <= 0 || T <= 0) fail(DSN_EINVAL, "dsn_decode: bad arguments");
    hipStream_t caller = (hipStream_t)stream;
    const int S = B * ctx->cfg.n_src;
    const long Lfull = (long)ctx->hop() * T;
    const long Lt = target_len > 0 ? target_len : Lfull;
    if (Lt > Lfull) fail(DSN_EINVAL, "target_len %ld > decoded length %ld", Lt, Lfull);
    const long esz = (long)S * ctx->cfg.latent_dim * T;
    float* eb = ctx->wsbuf<float>("dec_est", esz);
    hipStream_t st = ctx->enter(caller, 1);
    HIPCHK(hipMemcpyAsync(eb, est, sizeof(float) * esz, hipMemcpyDeviceToDevice, st));
    char key[64];
    snprintf(key, sizeof key, "dec:%d:%d", S, T);
    float* w = nullptr;
    ctx->run_graphed(key, st, [&](hipStream_t s2) { w = ctx->decode(eb, S, T, s2); });
    if (!w) w = ctx->wsbuf<float>("dec_wav", (long)S * Lfull);
    HIPCHK(hipMemcpy2DAsync(wav, sizeof(float) * Lt, w, sizeof(float) * Lfull, sizeof(float) * Lt, S,
                            hipMemcpyDeviceToDevice, st));
    ctx->leave(caller, st, 1);
  });
}

int dsn_encode(dsn_ctx* ctx, const float* mix, const float* vae_noise, uint64_t seed, float* y, int B, int L,
               void* stream) {
  return guarded(ctx, [&] {
    if (!mix || !y || B <= 0 || L < 0) fail(DSN_EINVAL, "dsn_encode: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int h = ctx->hop();
    const int T = dsn_latent_frames(ctx, L);
    const long Lp = (long)T * h;
    float* padded = ctx->wsbuf<float>("enc_wav", (long)B * Lp);
    HIPCHK(hipMemsetAsync(padded, 0, sizeof(float) * B * Lp, st));
    if (L > 0)
      HIPCHK(hipMemcpy2DAsync(padded, sizeof(float) * Lp, mix, sizeof(float) * L, sizeof(float) * L, B,
                              hipMemcpyDeviceToDevice, st));
    const long nsz = (long)B * ctx->cfg.latent_dim * T;
    if (!vae_noise) {
      float* nz = ctx->wsbuf<float>("enc_noise", nsz);
      launch_randn(nz, nsz, seed ^ 0x5851F42D4C957F2DULL, 0, st);
      vae_noise = nz;
    }
    ctx->encode(padded, vae_noise, y, B, (int)Lp, st);
    HIPCHK(hipGetLastError());
  });
}

int dsn_decode_chunked(dsn_ctx* ctx, const float* est, float* wav, int B, int T, int target_len, int chunk_size,
                       int overlap, void* stream) {
  return guarded(ctx, [&] {
    if (!est || !wav || B <= 0 || T <= 0) fail(DSN_EINVAL, "dsn_decode_chunked: bad arguments");
    hipStream_t caller = (hipStream_t)stream;
    const int S = B * ctx->cfg.n_src;
    const long Lfull = (long)ctx->hop() * T;
    const long Lt = target_len > 0 ? target_len : Lfull;
    if (Lt > Lfull) fail(DSN_EINVAL, "target_len %ld > decoded length %ld", Lt, Lfull);
    const long esz = (long)S * ctx->cfg.latent_dim * T;
    float* eb = ctx->wsbuf<float>("dec_est", esz);
    hipStream_t st = ctx->enter(caller, 1);
    HIPCHK(hipMemcpyAsync(eb, est, sizeof(float) * esz, hipMemcpyDeviceToDevice, st));
    float* w = ctx->decode_chunked(eb, S, T, chunk_size, overlap, st);
    HIPCHK(hipMemcpy2DAsync(wav, sizeof(float) * Lt, w, sizeof(float) * Lfull, sizeof(float) * Lt, S,
                            hipMemcpyDeviceToDevice, st));
    ctx->leave(caller, st, 1);
  });
}

int dsn_encode_chunked(dsn_ctx* ctx, const float* mix, const float* vae_noise, uint64_t seed, float* y, int B, int L,
                       int chunk_size, int overlap, void* stream) {
  return guarded(ctx, [&] {
    if (!mix || !y || B <= 0 || L < 0) fail(DSN_EINVAL, "dsn_encode_chunked: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int h = ctx->hop();
    const int T = dsn_latent_frames(ctx, L);
    const long Lp = (long)T * h;
    float* padded = ctx->wsbuf<float>("enc_wav", (long)B * Lp);
    HIPCHK(hipMemsetAsync(padded, 0, sizeof(float) * B * Lp, st));
    if (L > 0)
      HIPCHK(hipMemcpy2DAsync(padded, sizeof(float) * Lp, mix, sizeof(float) * L, sizeof(float) * L, B,
                              hipMemcpyDeviceToDevice, st));
    const long nsz = (long)B * ctx->cfg.latent_dim * T;
    if (!vae_noise) {
      float* nz = ctx->wsbuf<float>("enc_noise", nsz);
      launch_randn(nz, nsz, seed ^ 0x5851F42D4C957F2DULL, 0, st);
      vae_noise = nz;
    }
    float* enc = ctx->encode_chunked(padded, B, T, chunk_size, overlap, st);
    launch_vae_sample(enc, vae_noise, y, B, ctx->cfg.latent_dim, T, st);
    HIPCHK(hipGetLastError());
  });
}

int dsn_separate(dsn_ctx* ctx, const float* mix, const float* vae_noise, const float* noise, uint64_t seed,
                 float* wav, int B, int L, int target_len, int N, int corrector_steps, float snr, float t_eps,
                 int denoise, int* nfe_out, void* stream) {
  if (!ctx) return DSN_EINVAL;
  const int T = dsn_latent_frames(ctx, L);
  int rc = DSN_OK;
  float* y = nullptr;
  float* x = nullptr;
  rc = guarded(ctx, [&] {
    const long ysz = (long)B * ctx->cfg.latent_dim * T;
    y = ctx->wsbuf<float>("sep_y", ysz);
    x = ctx->wsbuf<float>("sep_x", ysz * ctx->cfg.n_src);
  });
  if (rc) return rc;
  if ((rc = dsn_encode(ctx, mix, vae_noise, seed, y, B, L, stream))) return rc;
  if ((rc = dsn_pc_sample(ctx, y, noise, seed + 1, x, B, T, N, corrector_steps, snr, t_eps, denoise, nfe_out, stream)))
    return rc;
  return dsn_decode(ctx, x, wav, B, T, target_len > 0 ? target_len : L, stream);
}

// Scale-invariant SDR with permutation-invariant assignment (the step right after the path in
// evaluate_latent.py:118-136, where the reference calls fast_bss_eval.si_bss_eval_sources with
// compute_permutation=True).  Device: all n x n correlation / energy sums; host: SI-SDR matrix
// (no mean removal, eps 1e-10) and the best of the n! <= 24 permutations.
int dsn_si_sdr_pit(dsn_ctx* ctx, const float* ref, const float* est, int B, int n, int L, float* si_sdr_out,
                   int* perm_out, void* stream) {
  return guarded(ctx, [&] {
    if (!ref || !est || B <= 0 || n <= 0 || n > 4 || L <= 0) fail(DSN_EINVAL, "dsn_si_sdr_pit: bad arguments (n <= 4)");
    hipStream_t st = (hipStream_t)stream;
    double* dd = ctx->wsbuf<double>("sisdr", (long)B * n * n * 3);
    launch_sisdr_dots(ref, est, B, n, L, dd, st);
    std::vector<double> h((size_t)B * n * n * 3);
    HIPCHK(hipMemcpyAsync(h.data(), dd, sizeof(double) * h.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<int> p(n);
    for (int b = 0; b < B; ++b) {
      // sdr[i][j]: est_j scored against ref_i
      double sdr[4][4];
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          const double* v = &h[(((size_t)b * n + i) * n + j) * 3];
          const double eps = 1e-10;
          const double alpha = v[0] / (v[1] + eps);
          const double tgt = alpha * alpha * v[1];
          const double noise = v[2] - 2.0 * alpha * v[0] + tgt;
          sdr[i][j] = 10.0 * log10((tgt + eps) / (noise + eps));
        }
      for (int i = 0; i < n; ++i) p[i] = i;
      double best = -1e300;
      std::vector<int> bestp = p;
      do {
        double s = 0;
        for (int i = 0; i < n; ++i) s += sdr[i][p[i]];
        s /= n;
        if (s > best) {
          best = s;
          bestp = p;
        }
      } while (std::next_permutation(p.begin(), p.end()));
      for (int i = 0; i < n; ++i) {
        if (si_sdr_out) si_sdr_out[(size_t)b * n + i] = (float)sdr[i][bestp[i]];
        if (perm_out) perm_out[(size_t)b * n + i] = bestp[i];
      }
    }
  });
}

// SI-SDR / SI-SIR / SI-SAR of every estimate against the references (the decomposition of bss_eval with a
// one-tap, i.e. scale-invariant, distortion filter; evaluate_latent.py:118-136 calls
// fast_bss_eval.si_bss_eval_sources(ref, est, zero_mean=False, compute_permutation=True, clamp_db=100)):
//   e_target = <est_j, ref_i> ref_i / |ref_i|^2          (projection on the matched reference)
//   P est_j  = projection of est_j on span{ref_0 .. ref_{n-1}}   (n x n Gram solve)
//   e_interf = P est_j - e_target,  e_artif = est_j - P est_j
//   SI-SDR = |e_target|^2 / |est_j - e_target|^2,  SI-SIR = |e_target|^2 / |e_interf|^2,
//   SI-SAR = |P est_j|^2 / |e_artif|^2
// Device: all inner products (two launches of the dots kernel: ref x est and ref x ref), fp64 accumulation;
// host: the n <= 4 Gram solves and the permutation (perm_by: 0 = best mean SI-SDR, 1 = best mean SI-SIR --
// bss_eval's convention, "order according to SIR" evaluate_latent.py:124).  Values clamped to +-clamp_db.
int dsn_si_bss_eval(dsn_ctx* ctx, const float* ref, const float* est, int B, int n, int L, int perm_by, float clamp_db,
                    float* si_sdr_out, float* si_sir_out, float* si_sar_out, int* perm_out, void* stream) {
  return guarded(ctx, [&] {
    if (!ref || !est || B <= 0 || n <= 0 || n > 4 || L <= 0 || perm_by < 0 || perm_by > 1)
      fail(DSN_EINVAL, "dsn_si_bss_eval: bad arguments (n <= 4, perm_by 0|1)");
    hipStream_t st = (hipStream_t)stream;
    const size_t cnt = (size_t)B * n * n * 3;
    double* dd = ctx->wsbuf<double>("sibss", (long)cnt * 2);
    launch_sisdr_dots(ref, est, B, n, L, dd, st);
    launch_sisdr_dots(ref, ref, B, n, L, dd + cnt, st);
    std::vector<double> h(cnt * 2);
    HIPCHK(hipMemcpyAsync(h.data(), dd, sizeof(double) * h.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const double clamp = clamp_db > 0 ? clamp_db : 1e30, tiny = 1e-300;
    auto db = [&](double num, double den) {
      double v = 10.0 * log10(std::max(num, tiny) / std::max(den, tiny));
      return std::min(clamp, std::max(-clamp, v));
    };
    std::vector<int> p(n);
    for (int b = 0; b < B; ++b) {
      double G[4][4], X[4][4], ee[4], sdr[4][4], sir[4][4], sar[4];
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          const double* v = &h[(((size_t)b * n + i) * n + j) * 3];
          X[i][j] = v[0];
          ee[j] = v[2];
          G[i][j] = h[cnt + (((size_t)b * n + i) * n + j) * 3];
        }
      for (int j = 0; j < n; ++j) {
        // solve G c = X[:, j] by Gaussian elimination with partial pivoting (fp64, n <= 4)
        double A[4][5];
        for (int r = 0; r < n; ++r) {
          for (int c = 0; c < n; ++c) A[r][c] = G[r][c];
          A[r][n] = X[r][j];
        }
        bool singular = false;
        for (int c = 0; c < n; ++c) {
          int piv = c;
          for (int r = c + 1; r < n; ++r)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
          if (fabs(A[piv][c]) < 1e-300) {
            singular = true;
            break;
          }
          if (piv != c)
            for (int k = 0; k <= n; ++k) std::swap(A[piv][k], A[c][k]);
          for (int r = c + 1; r < n; ++r) {
            const double f = A[r][c] / A[c][c];
            for (int k = c; k <= n; ++k) A[r][k] -= f * A[c][k];
          }
        }
        double cvec[4] = {0, 0, 0, 0}, pe = 0;
        if (!singular) {
          for (int r = n - 1; r >= 0; --r) {
            double acc = A[r][n];
            for (int k = r + 1; k < n; ++k) acc -= A[r][k] * cvec[k];
            cvec[r] = acc / A[r][r];
          }
          for (int k = 0; k < n; ++k) pe += cvec[k] * X[k][j];  // |P est_j|^2 = c^T G c = c^T x
        }
        pe = std::min(std::max(pe, 0.0), ee[j]);
        sar[j] = db(pe, ee[j] - pe);
        for (int i = 0; i < n; ++i) {
          const double et = G[i][i] > 0 ? X[i][j] * X[i][j] / G[i][i] : 0.0;  // |e_target|^2
          sdr[i][j] = db(et, ee[j] - et);
          sir[i][j] = db(et, pe - et);
        }
      }
      for (int i = 0; i < n; ++i) p[i] = i;
      double best = -1e300;
      std::vector<int> bestp = p;
      do {
        double sc = 0;
        for (int i = 0; i < n; ++i) sc += perm_by ? sir[i][p[i]] : sdr[i][p[i]];
        if (sc > best) {
          best = sc;
          bestp = p;
        }
      } while (std::next_permutation(p.begin(), p.end()));
      for (int i = 0; i < n; ++i) {
        const size_t o = (size_t)b * n + i;
        if (si_sdr_out) si_sdr_out[o] = (float)sdr[i][bestp[i]];
        if (si_sir_out) si_sir_out[o] = (float)sir[i][bestp[i]];
        if (si_sar_out) si_sar_out[o] = (float)sar[bestp[i]];
        if (perm_out) perm_out[o] = bestp[i];
      }
    }
  });
}

// Development hook: copy `count` floats of the named workspace buffer to host memory.
int dsn_debug_read(dsn_ctx* ctx, const char* name, float* host, int64_t count) {
  return guarded(ctx, [&] {
    auto it = ctx->ws.find(name);
    if (it == ctx->ws.end()) fail(DSN_EINVAL, "no workspace buffer '%s'", name);
    if ((size_t)count * sizeof(float) > it->second.second) fail(DSN_EINVAL, "buffer '%s' smaller than request", name);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(host, it->second.first, sizeof(float) * count, hipMemcpyDeviceToHost));
  });
}

int dsn_enable_graphs(dsn_ctx* ctx, int enable) {
  if (!ctx) return DSN_EINVAL;
  ctx->use_graphs = enable != 0;
  return DSN_OK;
}

int64_t dsn_workspace_bytes(const dsn_ctx* ctx) { return ctx ? ctx->ws_bytes() : 0; }

int dsn_profile_begin(dsn_ctx* ctx) {
  return guarded(ctx, [&] {
    for (auto& r : ctx->prof) {
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    ctx->prof.clear();
    ctx->profiling = true;
  });
}

int dsn_profile_end(dsn_ctx* ctx, double* gemm_ms, double* gemm_flops, int64_t* gemm_launches) {
  return guarded(ctx, [&] {
    HIPCHK(hipDeviceSynchronize());
    double ms = 0, fl = 0;
    int64_t n_gemm = 0;
    ctx->hbm_ms = ctx->hbm_bytes = 0;
    ctx->hbm_launches = 0;
    ctx->prof_rows.clear();
    std::map<std::string, size_t> row_of;
    for (auto& r : ctx->prof) {
      float t = 0;
      HIPCHK(hipEventElapsedTime(&t, r.a, r.b));
      if (r.gemm) {
        ms += t;
        fl += r.flops;
        ++n_gemm;
      }
      if (r.hbm_bound) {
        ctx->hbm_ms += t;
        ctx->hbm_bytes += r.hbm_bytes;
        ++ctx->hbm_launches;
      }
      auto it = row_of.find(r.tag);
      if (it == row_of.end()) {
        it = row_of.emplace(r.tag, ctx->prof_rows.size()).first;
        ctx->prof_rows.push_back(ProfRow());
        ctx->prof_rows.back().name = r.tag;
      }
      ProfRow& row = ctx->prof_rows[it->second];
      row.ms += t;
      row.flops += r.flops;
      row.bytes += r.hbm_bytes;
      ++row.launches;
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    if (gemm_ms) *gemm_ms = ms;
    if (gemm_flops) *gemm_flops = fl;
    if (gemm_launches) *gemm_launches = n_gemm;
    ctx->prof.clear();
    ctx->profiling = false;
  });
}

int dsn_profile_hbm(dsn_ctx* ctx, double* ms, double* bytes, int64_t* launches) {
  if (!ctx) return DSN_EINVAL;
  if (ms) *ms = ctx->hbm_ms;
  if (bytes) *bytes = ctx->hbm_bytes;
  if (launches) *launches = ctx->hbm_launches;
  return DSN_OK;
}

int dsn_profile_rows(dsn_ctx* ctx, int max_rows, char* names, double* ms, double* flops, double* bytes,
                     int64_t* launches) {
  if (!ctx) return DSN_EINVAL;
  const int n = (int)std::min<size_t>(ctx->prof_rows.size(), (size_t)std::max(0, max_rows));
  for (int i = 0; i < n; ++i) {
    const ProfRow& r = ctx->prof_rows[i];
    if (names) {
      strncpy(names + (size_t)i * DSN_PROFILE_NAME_LEN, r.name.c_str(), DSN_PROFILE_NAME_LEN - 1);
      names[(size_t)i * DSN_PROFILE_NAME_LEN + DSN_PROFILE_NAME_LEN - 1] = 0;
    }
    if (ms) ms[i] = r.ms;
    if (flops) flops[i] = r.flops;
    if (bytes) bytes[i] = r.bytes;
    if (launches) launches[i] = r.launches;
  }
  return (int)ctx->prof_rows.size();
}

int dsn_test_igemm(dsn_ctx* ctx, const float* a, const float* w, float* out, int B, int Lin, int Cin, int N, int taps,
                   int in_stride, int tap_dil, int in_pad, int rows_per_b, int panel_rows, int panel_bn, void* stream) {
  return guarded(ctx, [&] {
    hipStream_t st = (hipStream_t)stream;
    const int P = ctx->P, PL = ctx->PL;
    const long an = (long)B * Lin * Cin, wn = (long)N * taps * Cin;
    op16_t* ap = ctx->wsbuf<op16_t>("t_a", an * P);
    op16_t* wp = ctx->wsbuf<op16_t>("t_w", wn * P);
    launch_to_planes(a, ap, an, PL, an, st);
    launch_to_planes(w, wp, wn, PL, wn, st);
    Packed pk;
    pk.w = wp;
    pk.ps = wn;
    pk.N = N;
    pk.Cin = Cin;
    pk.taps = taps;
    pk.K = taps * Cin;
    GemmDesc d = ctx->base_desc(ap, an, pk, B, rows_per_b, Lin);
    d.in_stride = in_stride;
    d.tap_dil = tap_dil;
    d.in_pad = in_pad;
    d.out_f32 = out;
    if (panel_rows > 0) {
      d.panel_rows = panel_rows;
      hipError_t e = igemm_panel_launch(d, PL, panel_bn, st);
      if (e != hipSuccess) fail(DSN_EHIP, "panel launch: %s", hipGetErrorString(e));
    } else {
      ctx->run(d, st);
    }
    HIPCHK(hipGetLastError());
  });
}

// Development hook: time `iters` back-to-back launches of the implicit-GEMM kernel on random operands.
int dsn_bench_igemm(dsn_ctx* ctx, int B, int Lin, int Cin, int N, int taps, int tap_dil, int in_pad, int ksplit,
                    int variant, int iters, double* ms_out) {
  return guarded(ctx, [&] {
    const int P = ctx->P, PL = ctx->PL;
    const long an = (long)B * Lin * Cin, wn = (long)N * taps * Cin, on = (long)B * Lin * N;
    float* af = ctx->wsbuf<float>("b_af", an);
    float* wf = ctx->wsbuf<float>("b_wf", wn);
    op16_t* ap = ctx->wsbuf<op16_t>("b_a", an * P);
    op16_t* wp = ctx->wsbuf<op16_t>("b_w", wn * P);
    float* out = ctx->wsbuf<float>("b_o", on * (ksplit > 1 ? ksplit : 1));
    launch_randn(af, an, 1, 0, nullptr);
    launch_randn(wf, wn, 2, 0, nullptr);
    launch_to_planes(af, ap, an, PL, an, nullptr);
    launch_to_planes(wf, wp, wn, PL, wn, nullptr);
    Packed pk;
    pk.w = wp;
    pk.ps = wn;
    pk.N = N;
    pk.Cin = Cin;
    pk.taps = taps;
    pk.K = taps * Cin;
    GemmDesc d = ctx->base_desc(ap, an, pk, B, Lin, Lin);
    d.tap_dil = tap_dil;
    d.in_pad = in_pad;
    d.out_f32 = out;
    d.ksplit = ksplit;
    d.slab_stride = on;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    auto launch = [&] {
      d.m_fast = (variant & 0x40) ? 1 : 0;
      if (variant & 0x20) {
        d.panel_rows = (variant >> 8) & 0xfff;
        d.panel_wm = (variant & 0x80) ? 2 : 0;     // 8-wave variants; low bits then pick the ring (nst, BK 64 flag)
        if (variant & 0x80) {
          d.cfg_nst = variant & 0xf;
          d.cfg_bk = (variant & 0x10) ? 64 : 32;
        }
        hipError_t e2 = igemm_panel_launch(d, PL, (variant >> 20) & 0x3ff, nullptr);
        if (e2 != hipSuccess) fail(DSN_EHIP, "bench panel launch: %s", hipGetErrorString(e2));
        return;
      }
      hipError_t e = variant == 1 ? igemm_launch(d, PL, nullptr)
                                  : igemm2_launch_cfg(d, PL, (variant >> 8) & 0xfff, (variant >> 20) & 0x3ff,
                                                      variant & 0xf, (variant & 0x10) ? 64 : 32, nullptr);
      if (e != hipSuccess) fail(DSN_EHIP, "bench launch: %s", hipGetErrorString(e));
    };
    launch();
    launch();
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) launch();
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  });
}

}  // extern "C"
