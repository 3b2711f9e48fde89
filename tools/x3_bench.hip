// A/B/C... of dense-layer kernel variants on the layer shapes of the C2 step (tools only, not part of librgnn.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/x3_bench.hip -ldl -o tools/x3_bench.bin
//   tools/x3_bench.bin [-r rounds] [-s 0,1,7] variant [variant ...]
// A variant is  path/to/librgnn.so[:ENV=VALUE[:ENV=VALUE]]  (the variables are set around its calls only), e.g.
//   radargnn_amd/librgnn.so:RGNN_X3_NODMA=1   radargnn_amd/librgnn.so   tools/var/foo/librgnn.so
// Every library is dlopen'ed privately, so all variants run in ONE process on the same buffers: outputs are compared bit for
// bit with the first variant's, timings are taken in interleaved rounds (median reported).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../include/rgnn.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Shape { int64_t m; int k1, k2, n; int64_t subset; int stats; const char* name; };   // subset > 0: row_index launch over that many rows
struct Variant {
  std::string label;
  std::vector<std::pair<std::string, std::string>> env;
  int (*fwd)(const rgnn_linear_args*, rgnn_stream_t);
  int (*split)(const float*, const float*, int64_t, int32_t, int32_t, int32_t, void*, rgnn_stream_t);
  int32_t (*kp)(int32_t);
  int64_t (*panels)(int64_t);
  const char* (*err)(void);
  int64_t (*f16_bytes)(int32_t, int32_t);
  int (*split_f16)(const float*, const float*, int64_t, int32_t, int32_t, int32_t, void*, rgnn_stream_t);
  void (*env_reload)(void) = nullptr;                  // rgnn_env_reload: the library caches its environment reads
  int (*timing)(unsigned long long*, int) = nullptr;   // rgnn_debug_dma_timing of a -DRGNN_DMA_TIMING build
  bool f16 = false;      // variant spec carries F16=1: the f16x2 form (two f16 terms, three products) with host-computed bounds
};

int main(int argc, char** argv) {
  // (row-subset sizes: the C2 batch of bench.py has 133 516 nodes with edges and 58 484 without; the committed r02 x3_bench
  //  outputs up to r02_x3_bench_four_wave_form.txt were taken with 105 600 / 86 400, an early estimate)
  std::vector<Shape> shapes = {
      {192000, 224, 0, 464, 133516, 0, "Q   (rows with edges)"},
      {192000, 224, 464, 224, 133516, 1, "upd (rows with edges)"},
      {192000, 224, 0, 224, 58484, 1, "upd (isolated rows)"},
      {192000, 224, 464, 128, 133516, 1, "upd L3"},
      {192000, 128, 0, 272, 133516, 0, "Q L4"},
      {192000, 64, 0, 128, 0, 0, "emb 64->128 (dense)"},
      {192000, 128, 0, 224, 0, 0, "emb 128->224 (dense)"},
      {192000, 224, 464, 224, 0, 1, "upd dense"},
      {1000, 32, 16, 96, 0, 1, "small"},
      {777, 48, 0, 200, 500, 1, "small subset"},
  };
  int rounds = 7;
  bool permute = false;
  std::vector<int> pick;
  std::vector<Variant> vars;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-r")) { rounds = atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "-p")) { permute = true; continue; }
    if (!strcmp(argv[i], "-s")) { char* t = strtok(argv[++i], ","); while (t) { pick.push_back(atoi(t)); t = strtok(nullptr, ","); } continue; }
    Variant v;
    v.label = argv[i];
    std::string spec = argv[i], path = spec.substr(0, spec.find(':'));
    size_t pos = spec.find(':');
    while (pos != std::string::npos) {
      size_t nx = spec.find(':', pos + 1);
      std::string kv = spec.substr(pos + 1, nx == std::string::npos ? std::string::npos : nx - pos - 1);
      if (kv == "F16=1") v.f16 = true;
      else v.env.push_back({kv.substr(0, kv.find('=')), kv.substr(kv.find('=') + 1)});
      pos = nx;
    }
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen %s: %s\n", path.c_str(), dlerror()); return 1; }
    v.fwd = (decltype(v.fwd))dlsym(h, "rgnn_linear_fwd");
    v.split = (decltype(v.split))dlsym(h, "rgnn_linear_split_weights");
    v.kp = (decltype(v.kp))dlsym(h, "rgnn_linear_planes_kp");
    v.panels = (decltype(v.panels))dlsym(h, "rgnn_linear_stat_panels");
    v.err = (decltype(v.err))dlsym(h, "rgnn_last_error");
    v.f16_bytes = (decltype(v.f16_bytes))dlsym(h, "rgnn_linear_planes_f16_bytes");
    v.split_f16 = (decltype(v.split_f16))dlsym(h, "rgnn_linear_split_weights_f16");
    v.timing = (decltype(v.timing))dlsym(h, "rgnn_debug_dma_timing");
    v.env_reload = (decltype(v.env_reload))dlsym(h, "rgnn_env_reload");
    if (v.f16 && (!v.f16_bytes || !v.split_f16)) { printf("%s has no f16x2 entry points\n", path.c_str()); return 1; }
    vars.push_back(v);
  }
  if (vars.empty()) { printf("usage: %s [-r rounds] [-s shapes] lib.so[:ENV=V] ...\n", argv[0]); return 1; }
  if (pick.empty()) for (size_t i = 0; i < shapes.size(); i++) pick.push_back((int)i);
  const int nv = (int)vars.size();
  const int64_t skbytes = (int64_t)256 * 8 * 16 * 512 * 4 + 4096;
  void* skws;
  void* flush = nullptr;
  if (getenv("X3_FLUSH")) CK(hipMalloc(&flush, (size_t)1 << 30));
  CK(hipMalloc(&skws, skbytes));
  CK(hipMemset(skws, 0, skbytes));
  for (int si : pick) {
    const Shape s = shapes[si];
    const int K = s.k1 + s.k2;
    float *A1, *A2 = nullptr, *W, *b;
    std::vector<float*> out(nv), stats(nv);
    CK(hipMalloc(&A1, s.m * (size_t)s.k1 * 4));
    if (s.k2) CK(hipMalloc(&A2, s.m * (size_t)s.k2 * 4));
    CK(hipMalloc(&W, (size_t)s.n * K * 4));
    CK(hipMalloc(&b, s.n * 4));
    const size_t stat_n = vars[0].panels(s.m) * RGNN_STAT_ROWS * (size_t)s.n;
    for (int v = 0; v < nv; v++) { CK(hipMalloc(&out[v], s.m * (size_t)s.n * 4)); CK(hipMalloc(&stats[v], stat_n * 4)); }
    srand(1234);
    std::vector<float> h(s.m * (size_t)std::max(s.k1, s.k2));
    for (auto& v : h) v = ((float)rand() / RAND_MAX - 0.5f) * 4.f;
    CK(hipMemcpy(A1, h.data(), s.m * (size_t)s.k1 * 4, hipMemcpyHostToDevice));
    const int64_t ref_rows = std::min<int64_t>(s.m, 512);                  // float64 reference on the first rows (accuracy column)
    std::vector<float> hA1(h.begin(), h.begin() + ref_rows * s.k1), hA2;
    for (auto& v : h) v = ((float)rand() / RAND_MAX - 0.3f) * 2.f;
    if (s.k2) { CK(hipMemcpy(A2, h.data(), s.m * (size_t)s.k2 * 4, hipMemcpyHostToDevice)); hA2.assign(h.begin(), h.begin() + ref_rows * s.k2); }
    float bounds_h[2] = {2.f, 1.4f};                                       // |A1| <= 2, |A2| <= 1.4 by construction
    float* bounds_d; CK(hipMalloc(&bounds_d, 2 * RGNN_BOUND_SLOTS * 4)); CK(hipMemset(bounds_d, 0, 2 * RGNN_BOUND_SLOTS * 4));
    CK(hipMemcpy(bounds_d, bounds_h, 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bounds_d + RGNN_BOUND_SLOTS, bounds_h + 1, 4, hipMemcpyHostToDevice));
    float* amax_d; CK(hipMalloc(&amax_d, RGNN_BOUND_SLOTS * 4)); CK(hipMemset(amax_d, 0, RGNN_BOUND_SLOTS * 4));
    std::vector<float> hw((size_t)s.n * K), hb(s.n);
    for (auto& v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
    for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), s.n * 4, hipMemcpyHostToDevice));
    int32_t* ridx = nullptr; int64_t* mdev = nullptr;
    if (s.subset) {
      std::vector<int32_t> idx(s.m);
      for (int64_t i = 0; i < s.m; i++) idx[i] = (int32_t)i;
      for (int64_t i = s.m - 1; i > 0; i--) { int64_t j = rand() % (i + 1); std::swap(idx[i], idx[j]); }
      std::sort(idx.begin(), idx.begin() + s.subset);
      // the model's row lists are in grid-cell visiting order: ascending by frame, scattered inside a frame (-p)
      if (permute)
        for (int64_t f0 = 0; f0 < s.subset; f0 += 1650)
          for (int64_t i = std::min<int64_t>(s.subset, f0 + 1650) - 1; i > f0; i--) std::swap(idx[i], idx[f0 + rand() % (i - f0 + 1)]);
      CK(hipMalloc(&ridx, s.m * 4)); CK(hipMalloc(&mdev, 8));
      CK(hipMemcpy(ridx, idx.data(), s.m * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(mdev, &s.subset, 8, hipMemcpyHostToDevice));
    }
    const int kp = vars[0].kp(K);
    void* planes;
    CK(hipMalloc(&planes, (size_t)3 * s.n * kp * 2));
    vars[0].split(W, nullptr, K, s.n, s.n, K, planes, nullptr);
    std::vector<void*> planes16(nv, nullptr);
    for (int v = 0; v < nv; v++)
      if (vars[v].f16) {
        CK(hipMalloc(&planes16[v], vars[v].f16_bytes(s.n, K)));
        if (vars[v].split_f16(W, nullptr, K, s.n, s.n, K, planes16[v], nullptr)) { printf("split_f16 failed: %s\n", vars[v].err()); exit(1); }
      }
    auto run = [&](int v) {
      for (auto& e : vars[v].env) setenv(e.first.c_str(), e.second.c_str(), 1);
      if (vars[v].env_reload) vars[v].env_reload();
      rgnn_linear_args a = {};
      a.A1 = A1; a.lda1 = s.k1; a.k1 = s.k1; a.A2 = A2; a.lda2 = s.k2; a.k2 = s.k2;
      a.W1 = W; a.W2 = nullptr; a.ldw = K; a.w_split = s.n; a.bias1 = b; a.out = out[v]; a.ldo = s.n; a.m = s.m; a.n = s.n;
      a.relu_out = 1; a.col_stats = (s.stats && !getenv("X3_NO_STATS")) ? stats[v] : nullptr; a.W_planes = planes; a.w_planes_kp = kp;
      a.row_index = ridx; a.m_dev = mdev;
      if (!getenv("X3_NO_SK")) { a.splitk_ws = skws; a.splitk_ws_bytes = skbytes; }
      if (vars[v].f16) { a.W_planes_f16 = planes16[v]; a.a1_bound = bounds_d; a.a2_bound = bounds_d + RGNN_BOUND_SLOTS; }
      if (getenv("X3_ABSMAX")) a.out_absmax = amax_d;
      int rc = vars[v].fwd(&a, nullptr);
      for (auto& e : vars[v].env) unsetenv(e.first.c_str());
      if (vars[v].env_reload && !vars[v].env.empty()) vars[v].env_reload();
      if (rc) { printf("rgnn_linear_fwd failed: %s\n", vars[v].err()); exit(1); }
    };
    const double rows = s.subset ? s.subset : s.m;
    const double fl = 2.0 * rows * K * s.n;
    printf("%-22s M=%6.0f K=%3d N=%3d\n", s.name, rows, K, s.n);
    std::vector<float> o0(s.m * (size_t)s.n), o1(o0.size()), s0(stat_n), s1(stat_n);
    std::vector<size_t> bad(nv, 0), bad_s(nv, 0);
    std::vector<double> err64(nv, 0.0);
    // float64 reference of the first rows (dense launches: tile row = matrix row; subsets: the rows the list names first)
    std::vector<int32_t> ref_idx(ref_rows);
    for (int64_t i = 0; i < ref_rows; i++) ref_idx[i] = (int32_t)i;
    std::vector<double> ref(ref_rows * (size_t)s.n);
    double ref_max = 0.0;
    for (int64_t r = 0; r < ref_rows; r++)
      for (int c = 0; c < s.n; c++) {
        double acc = hb[c];
        for (int k = 0; k < s.k1; k++) acc += (double)hA1[r * s.k1 + k] * hw[(size_t)c * K + k];
        for (int k = 0; k < s.k2; k++) acc += (double)hA2[r * s.k2 + k] * hw[(size_t)c * K + s.k1 + k];
        acc = acc > 0 ? acc : 0;
        ref[r * s.n + c] = acc;
        ref_max = std::max(ref_max, fabs(acc));
      }
    std::vector<char> row_live(s.m, s.subset ? 0 : 1);
    for (int v = 0; v < nv; v++) {
      CK(hipMemset(out[v], 0, s.m * (size_t)s.n * 4)); CK(hipMemset(stats[v], 0, stat_n * 4));
      run(v);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy((v ? o1 : o0).data(), out[v], o0.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy((v ? s1 : s0).data(), stats[v], stat_n * 4, hipMemcpyDeviceToHost));
      {
        const std::vector<float>& o = v ? o1 : o0;
        if (s.subset && !v) {                         // which of the first rows does the list contain?
          std::vector<int32_t> idx(s.m);
          CK(hipMemcpy(idx.data(), ridx, s.m * 4, hipMemcpyDeviceToHost));
          for (int64_t i = 0; i < s.subset; i++) row_live[idx[i]] = 1;
        }
        double e = 0.0;
        for (int64_t r = 0; r < ref_rows; r++)
          if (row_live[r])
            for (int c = 0; c < s.n; c++) e = std::max(e, fabs((double)o[r * s.n + c] - ref[r * s.n + c]));
        err64[v] = e / ref_max;
      }
      if (!v) continue;
      for (size_t i = 0; i < o0.size(); i++) if (memcmp(&o0[i], &o1[i], 4)) bad[v]++;
      const int64_t live_panels = vars[0].panels(s.subset ? s.subset : s.m);   // (partial sums are grouped differently: to rounding)
      if (s.stats)
        for (size_t i = 0; i < (size_t)live_panels * 2 * s.n; i++)
          if (fabs((double)s0[i] - s1[i]) / (fabs((double)s0[i]) + 1e-3) > 1e-4) bad_s[v]++;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> tms(nv);
    for (int r = 0; r < rounds; r++)
      for (int v = 0; v < nv; v++) {
        run(v);
        if (flush) {                               // X3_FLUSH=1: every timed launch starts with cold L2 / MALL (a 1 GiB memset in front)
          float tot = 0.f;
          for (int i = 0; i < 5; i++) {
            CK(hipMemsetAsync(flush, i, (size_t)1 << 30, nullptr));
            CK(hipEventRecord(e0, nullptr));
            run(v);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            tot += ms;
          }
          tms[v].push_back(tot / 5);
          continue;
        }
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 5; i++) run(v);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        tms[v].push_back(ms / 5);
      }
    for (int v = 0; v < nv; v++) {
      if (vars[v].timing) {                        // variant built with -DRGNN_DMA_TIMING: per-wave s_memtime sums of one launch
        std::vector<unsigned long long> tt(2048 * 8);
        vars[v].timing(nullptr, 1);
        run(v);
        CK(hipDeviceSynchronize());
        vars[v].timing(tt.data(), 0);
        double sum[8] = {0}; int waves = 0; double mx[8] = {0};
        for (int w = 0; w < 2048; w++) {
          if (!tt[w * 8 + 5]) continue;
          waves++;
          for (int i = 0; i < 8; i++) { sum[i] += (double)tt[w * 8 + i]; mx[i] = std::max(mx[i], (double)tt[w * 8 + i]); }
        }
        if (getenv("X3_TIMING_WAVES"))
          for (int w = 0; w < 16; w++) {
            const double n = (double)tt[w * 8 + 5];
            if (n > 0) printf("        wg %d wave %d: steps %.0f  wait %.0f  B1 %.0f  load half %.0f (own %.0f)  matrix half %.0f  epilogue %.0f  kernel %.0f\n", w / 8, w % 8, n, tt[w * 8] / n,
                              tt[w * 8 + 1] / n, tt[w * 8 + 2] / n, tt[w * 8 + 7] / n, tt[w * 8 + 3] / n, (double)tt[w * 8 + 4], (double)tt[w * 8 + 6]);
          }
        const double st = sum[5] / waves;
        printf("      timing (%d waves, %.1f k-steps each; cycles per k-step): wait %.0f  barrier %.0f  load half %.0f  matrix half %.0f | per wave: k-loop %.0f  epilogue issue %.0f  drain | own load work per step %.0f  whole kernel %.0f (max %.0f) cycles\n",
               waves, st, sum[0] / sum[5], sum[1] / sum[5], sum[2] / sum[5], sum[3] / sum[5], (sum[0] + sum[1] + sum[2] + sum[3]) / waves, sum[4] / waves, sum[7] / sum[5], sum[6] / waves, mx[6]);
      }
      std::sort(tms[v].begin(), tms[v].end());
      const double t = tms[v][rounds / 2], tmin = tms[v][0];
      printf("    %-58s %7.1f us (min %7.1f) %6.1f TF  %4.1f%% of bf16 peak", vars[v].label.c_str(), t * 1e3, tmin * 1e3, fl / t / 1e9,
             6 * fl / t / 1e9 / 2500 * 100);
      printf("  err64 %.2e", err64[v]);
      if (v) printf("  x%.3f vs first | mismatches out %zu stats %zu", tms[0][rounds / 2] / t, bad[v], bad_s[v]);
      printf("\n");
    }
    hipFree(A1); hipFree(A2); hipFree(W); hipFree(b); hipFree(planes); hipFree(ridx); hipFree(mdev); hipFree(bounds_d); hipFree(amax_d);
    for (int v = 0; v < nv; v++) hipFree(planes16[v]);
    for (int v = 0; v < nv; v++) { hipFree(out[v]); hipFree(stats[v]); }
  }
  return 0;
}
