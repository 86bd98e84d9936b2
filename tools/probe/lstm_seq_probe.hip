// Stand-alone probe of the persistent LSTM recurrence (not part of the library): compiles csrc/lstm_seq.hip with in-kernel
// timestamps (VMMT_SEQ_PROBE) and prints where a step spends its time, on an otherwise idle chip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ivariational_mmt_amd/csrc tools/probe/lstm_seq_probe.hip -o gpurun_out/lstm_seq_probe
#define VMMT_SEQ_PROBE 1
#include "../../variational_mmt_amd/csrc/lstm.hip"
#include "../../variational_mmt_amd/csrc/lstm_seq.hip"
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 512, ndir = argc > 2 ? atoi(argv[2]) : 1, B = argc > 3 ? atoi(argv[3]) : 256, T = argc > 4 ? atoi(argv[4]) : 20;   // e.g. 256 2 21 256 = encoder_tgt of the conditional model
  const long M = (long)T * B;
  void *hout, *whh, *gs, *xchg; float *gx, *cc; unsigned* sync; vmmt_lstm_dir_fwd* ddev;
  CK(hipMalloc(&hout, (M + B) * H * 2 * ndir)); CK(hipMalloc(&whh, (long)4 * H * H * 2 * ndir)); CK(hipMalloc(&gs, M * 4 * H * 2 * ndir));
  CK(hipMalloc(&gx, M * 4 * H * 4 * ndir)); CK(hipMalloc(&cc, (M + B) * H * 4 * ndir));
  CK(hipMemset(hout, 0, (M + B) * H * 2 * ndir)); CK(hipMemset(whh, 0, (long)4 * H * H * 2 * ndir)); CK(hipMemset(gx, 0, M * 4 * H * 4 * ndir));
  CK(hipMemset(cc, 0, (M + B) * H * 4 * ndir));
  const long xb = vmmt_lstm_seq_xchg_bytes(ndir, B, H);
  CK(hipMalloc(&xchg, xb)); CK(hipMemset(xchg, 0, xb)); CK(hipMalloc(&sync, 4 * vmmt_lstm_seq_sync_words())); CK(hipMemset(sync, 0, 4 * vmmt_lstm_seq_sync_words()));
  std::vector<vmmt_lstm_dir_fwd> d((size_t)T * ndir);
  for (int t = 0; t < T; ++t)
    for (int k = 0; k < ndir; ++k) {
      vmmt_lstm_dir_fwd& x = d[(size_t)t * ndir + k];
      x = vmmt_lstm_dir_fwd{};
      char* hk = (char*)hout + (long)k * (M + B) * H * 2; float* ck = cc + (long)k * (M + B) * H;
      x.h_prev = hk + (long)t * B * H * 2; x.ld_hprev = H; x.c_prev = ck + (long)t * B * H; x.ld_cprev = H;
      x.w_hh = (char*)whh + (long)k * 4 * H * H * 2; x.ld_w = H;
      x.gx = gx + ((long)k * M + (long)t * B) * 4 * H; x.ld_gx = 4 * H;
      x.gates = (char*)gs + ((long)k * M + (long)t * B) * 4 * H * 2; x.ld_gates = 4 * H;
      x.c_out = ck + (long)(t + 1) * B * H; x.ld_c = H; x.h_out = hk + (long)(t + 1) * B * H * 2; x.ld_h = H;
      x.t = t;
    }
  CK(hipMalloc(&ddev, d.size() * sizeof(d[0]))); CK(hipMemcpy(ddev, d.data(), d.size() * sizeof(d[0]), hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode) {
    auto run = [&](int reps) {
      for (int r = 0; r < reps; ++r) {
        int rc = mode ? vmmt_lstm_seq_fwd(VMMT_BF16, ndir, T, d.data(), ddev, nullptr, B, H, sync, xchg, st)
                      : vmmt_lstm_chain_fwd(VMMT_BF16, ndir, T, d.data(), nullptr, B, H, st);
        if (rc) { printf("rc %d\n", rc); return; }
      }
    };
    run(3); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); run(20); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s H=%d ndir=%d: %.2f us per step (%.1f us per %d-step sequence)\n", mode ? "persistent" : "per-step  ", H, ndir, ms * 1e3 / (20 * T),
           ms * 1e3 / 20, T);
  }
  unsigned sw[4]; CK(hipMemcpy(sw, sync, 16, hipMemcpyDeviceToHost));
  printf("sync: epoch %u finish %u err 0x%x\n", sw[0], sw[1], sw[2]);
  std::vector<unsigned> xc(512);
  CK(hipMemcpyFromSymbol(xc.data(), HIP_SYMBOL(vmmt::vmmt_seq_xcc), xc.size() * 4));
  printf("XCC id of blocks 0..31:");
  for (int b = 0; b < 32; ++b) printf(" %u", xc[b] & 15u);
  int fast = 0, nb = ndir * ((B + 31) / 32) * (H / 16);
  for (int b = 0; b < nb; ++b) fast += (xc[b] >> 8) == 1;
  printf("\nblocks whose group uses the same-XCD transport: %d of %d\n", fast, nb);
  std::vector<unsigned long long> ts(8 * 64 * 8);
  CK(hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(vmmt::vmmt_seq_ts), ts.size() * 8));
  const char* names[8] = {"top", "polled", "swept", "mfma", "fold-bar", "cell", "tile-bar", "published"};
  for (int b = 0; b < 2; ++b)
    for (int t = 1; t < 6; ++t) {
      const unsigned long long* x = &ts[(b * 64 + t) * 8];
      printf("block %d step %d:", b, t);
      for (int i = 1; i < 8; ++i) printf(" %s +%.2f", names[i], (double)(x[i] - x[0]) * 0.01);
      printf("   | step period %.2f us\n", (double)(x[0] - ts[(b * 64 + t - 1) * 8]) * 0.01);
    }
  // ---------------- backward ----------------
  {
    void *dg, *whhT, *dha, *gates; float *c, *cn, *dcc; vmmt_lstm_dir_bwd* bdev; void* xb2;
    CK(hipMalloc(&dg, M * 4 * H * 2 * ndir)); CK(hipMalloc(&whhT, (long)H * 4 * H * 2 * ndir)); CK(hipMalloc(&dha, M * H * 2 * ndir));
    CK(hipMalloc(&gates, M * 4 * H * 2 * ndir)); CK(hipMalloc(&c, M * H * 4 * ndir)); CK(hipMalloc(&cn, (long)B * H * 4 * ndir));
    CK(hipMalloc(&dcc, (long)B * H * 4 * ndir));
    CK(hipMemset(dg, 0, M * 4 * H * 2 * ndir)); CK(hipMemset(whhT, 0, (long)H * 4 * H * 2 * ndir)); CK(hipMemset(dha, 0, M * H * 2 * ndir));
    CK(hipMemset(gates, 0, M * 4 * H * 2 * ndir)); CK(hipMemset(c, 0, M * H * 4 * ndir)); CK(hipMemset(cn, 0, (long)B * H * 4 * ndir));
    CK(hipMemset(dcc, 0, (long)B * H * 4 * ndir));
    const long xbb = vmmt_lstm_seq_xchg_bytes_bwd(ndir, B, H);
    CK(hipMalloc(&xb2, xbb)); CK(hipMemset(xb2, 0, xbb)); CK(hipMemset(sync, 0, 4 * vmmt_lstm_seq_sync_words()));
    std::vector<vmmt_lstm_dir_bwd> db((size_t)T * ndir);
    for (int step = 0; step < T; ++step)
      for (int k = 0; k < ndir; ++k) {
        const int t = T - 1 - step;
        vmmt_lstm_dir_bwd& x = db[(size_t)step * ndir + k];
        x = vmmt_lstm_dir_bwd{};
        char* dgk = (char*)dg + (long)k * M * 4 * H * 2;
        x.dgates_next = step > 0 ? dgk + (long)(t + 1) * B * 4 * H * 2 : nullptr; x.ld_dgn = 4 * H;
        x.w_hh_t = (char*)whhT + (long)k * H * 4 * H * 2; x.ld_wt = 4 * H;
        x.dh_above = (char*)dha + ((long)k * M + (long)t * B) * H * 2; x.ld_dha = H;
        x.gates = (char*)gates + ((long)k * M + (long)t * B) * 4 * H * 2; x.ld_gates = 4 * H;
        x.c_t = c + ((long)k * M + (long)t * B) * H; x.ld_ct = H;
        x.c_prev = t > 0 ? c + ((long)k * M + (long)(t - 1) * B) * H : cn + (long)k * B * H; x.ld_cp = H;
        x.dc_carry = dcc + (long)k * B * H; x.ld_dcc = H;
        x.dgates_out = dgk + (long)t * B * 4 * H * 2; x.ld_dgo = 4 * H;
        x.t = t; x.inject = 0;
      }
    CK(hipMalloc(&bdev, db.size() * sizeof(db[0]))); CK(hipMemcpy(bdev, db.data(), db.size() * sizeof(db[0]), hipMemcpyHostToDevice));
    for (int mode = 0; mode < 2; ++mode) {
      auto run = [&](int reps) {
        for (int r = 0; r < reps; ++r) {
          int rc = mode ? vmmt_lstm_seq_bwd(VMMT_BF16, ndir, T, db.data(), bdev, nullptr, B, H, 0, sync, xb2, st)
                        : vmmt_lstm_chain_bwd(VMMT_BF16, ndir, T, db.data(), nullptr, B, H, 0, st);
          if (rc) { printf("rc %d\n", rc); return; }
        }
      };
      run(3); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st)); run(20); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("BWD %s H=%d ndir=%d: %.2f us per step (%.1f us per %d-step sequence)\n", mode ? "persistent" : "per-step  ", H, ndir,
             ms * 1e3 / (20 * T), ms * 1e3 / 20, T);
    }
    CK(hipMemcpy(sw, sync, 16, hipMemcpyDeviceToHost));
    printf("sync: epoch %u finish %u err 0x%x\n", sw[0], sw[1], sw[2]);
    CK(hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(vmmt::vmmt_seq_ts), ts.size() * 8));
    const char* nb2[8] = {"top", "polled", "swept+mfma", "fold-wr", "fold-bar", "cell", "tile-bar", "published"};
    for (int t = 1; t < 5; ++t) {
      const unsigned long long* x = &ts[(0 * 64 + t) * 8];
      printf("bwd block 0 step %d:", t);
      for (int i = 1; i < 8; ++i) printf(" %s +%.2f", nb2[i], (double)(x[i] - x[0]) * 0.01);
      printf("   | step period %.2f us\n", (double)(x[0] - ts[(0 * 64 + t - 1) * 8]) * 0.01);
    }
  }
  return 0;
}
