// Stand-alone probe of the bf16 LSTM step kernels (not part of the library): compiles csrc/lstm.hip with in-kernel
// timestamps (VMMT_PROBE) and prints where a backward / forward step spends its time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ivariational_mmt_amd/csrc tools/probe/lstm_probe.hip -o gpurun_out/lstm_probe
#define VMMT_PROBE 1
#include "../../variational_mmt_amd/csrc/lstm.hip"
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int B = 256, H = argc > 1 ? atoi(argv[1]) : 512, T = 20, ndir = argc > 2 ? atoi(argv[2]) : 1;
  const long M = (long)T * B;
  void *dg, *whhT, *dha, *gates; float *c, *cn, *dcc;
  CK(hipMalloc(&dg, M * 4 * H * 2 * ndir)); CK(hipMalloc(&whhT, (long)H * 4 * H * 2 * ndir)); CK(hipMalloc(&dha, M * H * 2 * ndir));
  CK(hipMalloc(&gates, M * 4 * H * 2 * ndir)); CK(hipMalloc(&c, M * H * 4 * ndir)); CK(hipMalloc(&cn, (long)B * H * 4 * ndir));
  CK(hipMalloc(&dcc, (long)B * H * 4 * ndir));
  CK(hipMemset(dg, 0, M * 4 * H * 2 * ndir)); CK(hipMemset(whhT, 0, (long)H * 4 * H * 2 * ndir)); CK(hipMemset(dha, 0, M * H * 2 * ndir));
  CK(hipMemset(gates, 0, M * 4 * H * 2 * ndir)); CK(hipMemset(c, 0, M * H * 4 * ndir)); CK(hipMemset(cn, 0, (long)B * H * 4 * ndir));
  CK(hipMemset(dcc, 0, (long)B * H * 4 * ndir));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int reps) {
    for (int r = 0; r < reps; ++r)
      for (int t = T - 1; t >= 0; --t) {
        vmmt_lstm_dir_bwd d[2] = {};
        for (int k = 0; k < ndir; ++k) {
          char* dgk = (char*)dg + (long)k * M * 4 * H * 2;
          d[k].dgates_next = t < T - 1 ? dgk + (long)(t + 1) * B * 4 * H * 2 : nullptr; d[k].ld_dgn = 4 * H;
          d[k].w_hh_t = (char*)whhT + (long)k * H * 4 * H * 2; d[k].ld_wt = 4 * H;
          d[k].dh_above = (char*)dha + ((long)k * M + (long)t * B) * H * 2; d[k].ld_dha = H;
          d[k].gates = (char*)gates + ((long)k * M + (long)t * B) * 4 * H * 2; d[k].ld_gates = 4 * H;
          d[k].c_t = c + ((long)k * M + (long)t * B) * H; d[k].ld_ct = H;
          d[k].c_prev = t > 0 ? c + ((long)k * M + (long)(t - 1) * B) * H : cn + (long)k * B * H; d[k].ld_cp = H;
          d[k].dc_carry = dcc + (long)k * B * H; d[k].ld_dcc = H;
          d[k].dgates_out = dgk + (long)t * B * 4 * H * 2; d[k].ld_dgo = 4 * H;
          d[k].t = t; d[k].inject = 0;
        }
        int rc = vmmt_lstm_step_bwd(VMMT_BF16, ndir, d, nullptr, B, H, 0, st);
        if (rc) { printf("rc %d\n", rc); return; }
      }
  };
  run(2); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st)); run(10); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("bwd H=%d ndir=%d: %.2f us per step\n", H, ndir, ms * 1e3 / (10 * T));
  std::vector<unsigned long long> ts(64 * 16);
  CK(hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(vmmt::vmmt_probe_ts), ts.size() * 8));
  const char* names[8] = {"start", "issued1", "landed1", "mfma1", "issued2", "landed2", "mfma2", "end"};
  for (int b = 0; b < 8; ++b) {
    printf("block x=%d:", b);
    for (int i = 1; i < 8; ++i) printf(" %s +%.2f", names[i], (double)(ts[b * 16 + i] - ts[b * 16]) * 0.01);
    printf("  (us since start; start skew vs block0 %.2f)\n", (double)((long long)(ts[b * 16] - ts[0])) * 0.01);
  }
  // ---------------- forward ----------------
  {
    void *hout, *whh, *gs; float *gx, *cc;
    CK(hipMalloc(&hout, (M + B) * H * 2 * ndir)); CK(hipMalloc(&whh, (long)4 * H * H * 2 * ndir)); CK(hipMalloc(&gs, M * 4 * H * 2 * ndir));
    CK(hipMalloc(&gx, M * 4 * H * 4 * ndir)); CK(hipMalloc(&cc, (M + B) * H * 4 * ndir));
    CK(hipMemset(hout, 0, (M + B) * H * 2 * ndir)); CK(hipMemset(whh, 0, (long)4 * H * H * 2 * ndir)); CK(hipMemset(gx, 0, M * 4 * H * 4 * ndir));
    CK(hipMemset(cc, 0, (M + B) * H * 4 * ndir));
    auto runf = [&](int reps) {
      for (int r = 0; r < reps; ++r)
        for (int t = 0; t < T; ++t) {
          vmmt_lstm_dir_fwd d[2] = {};
          for (int k = 0; k < ndir; ++k) {
            char* hk = (char*)hout + (long)k * (M + B) * H * 2; float* ck = cc + (long)k * (M + B) * H;
            d[k].h_prev = hk + (long)t * B * H * 2; d[k].ld_hprev = H; d[k].c_prev = ck + (long)t * B * H; d[k].ld_cprev = H;
            d[k].w_hh = (char*)whh + (long)k * 4 * H * H * 2; d[k].ld_w = H;
            d[k].gx = gx + ((long)k * M + (long)t * B) * 4 * H; d[k].ld_gx = 4 * H;
            d[k].gates = (char*)gs + ((long)k * M + (long)t * B) * 4 * H * 2; d[k].ld_gates = 4 * H;
            d[k].c_out = ck + (long)(t + 1) * B * H; d[k].ld_c = H; d[k].h_out = hk + (long)(t + 1) * B * H * 2; d[k].ld_h = H;
            d[k].t = t;
          }
          int rc = vmmt_lstm_step_fwd(VMMT_BF16, ndir, d, nullptr, B, H, st);
          if (rc) { printf("rc %d\n", rc); return; }
        }
    };
    runf(2); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); runf(10); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("fwd H=%d ndir=%d: %.2f us per step\n", H, ndir, ms * 1e3 / (10 * T));
    CK(hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(vmmt::vmmt_probe_ts), ts.size() * 8));
    const char* nf[8] = {"start", "issued", "landed", "mfma", "-", "-", "-", "end"};
    for (int b = 0; b < 8; ++b) {
      printf("block x=%d:", b);
      for (int i : {1, 2, 3, 7}) printf(" %s +%.2f", nf[i], (double)(ts[b * 16 + i] - ts[b * 16]) * 0.01);
      printf("\n");
    }
  }
  return 0;
}
