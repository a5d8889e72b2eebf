// C-ABI plumbing: error reporting and build identification (see include/gdmae_hip.h).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void gd_set_error(int code, const char* file, int line, const char* msg) {
  snprintf(g_err, sizeof(g_err), "gdmae_hip error %d at %s:%d: %s", code, file, line, msg ? msg : "");
}

extern "C" const char* gdmae_last_error(void) { return g_err; }

extern "C" int gdmae_abi_version(void) { return 1; }

extern "C" const char* gdmae_target_arch(void) { return "gfx950"; }

// ------------------------------------------------------------------------------------------
// measurement slots (common.h GdTimed)
// ------------------------------------------------------------------------------------------
#include <vector>
int g_gd_timing_on = 0;
namespace {
struct TimedCall {
  hipEvent_t a, b;
  double bytes, flops, side;
};
std::vector<TimedCall> g_timed[GD_T_SLOTS];
const char* const kSlotNames[GD_T_SLOTS] = {"k_win_attn_fwd", "k_win_attn_bwd", "k_tok_gemm", "k_dw_grouped", "k_conv3x3_tiles",
                                            "k_conv_grad_taps", "k_spconv_fwd", "k_spconv_bwd", "k_dec_conv_bwd", "k_vfe",
                                            "k_plan", "k_layer_tail", "k_ffn", "k_rows_gemm", "", ""};
struct Pending {
  int slot;
  TimedCall tc;
};
}  // namespace
void* gd_timing_begin(int slot, hipStream_t st) {
  if (slot < 0 || slot >= GD_T_SLOTS) return nullptr;
  Pending* p = new Pending;
  p->slot = slot;
  if (hipEventCreate(&p->tc.a) != hipSuccess || hipEventCreate(&p->tc.b) != hipSuccess) {
    delete p;
    return nullptr;
  }
  (void)hipEventRecord(p->tc.a, st);
  return p;
}
void gd_timing_end(void* handle, hipStream_t st, double bytes, double flops, double side) {
  Pending* p = (Pending*)handle;
  (void)hipEventRecord(p->tc.b, st);
  p->tc.bytes = bytes;
  p->tc.flops = flops;
  p->tc.side = side;
  g_timed[p->slot].push_back(p->tc);
  delete p;
}
// on != 0: start collecting (earlier records dropped); 0: stop (records stay readable)
extern "C" int gdmae_kernel_timing(int on) {
  if (on) {
    for (int w = 0; w < GD_T_SLOTS; ++w) {
      for (auto& t : g_timed[w]) {
        (void)hipEventDestroy(t.a);
        (void)hipEventDestroy(t.b);
      }
      g_timed[w].clear();
    }
  }
  g_gd_timing_on = on;
  return 0;
}
extern "C" int gdmae_kernel_timing_slots(void) { return GD_T_SLOTS; }
extern "C" const char* gdmae_kernel_timing_name(int slot) { return (slot >= 0 && slot < GD_T_SLOTS) ? kSlotNames[slot] : ""; }
// summed milliseconds, number of bracketed calls, summed algorithmic bytes / flops of a slot (synchronises with its events)
extern "C" int gdmae_kernel_timing_read(int slot, double* total_ms, long long* calls, double* bytes, double* flops) {
  GD_REQUIRE(slot >= 0 && slot < GD_T_SLOTS, "kernel timing: no such slot");
  double tot = 0.0, by = 0.0, fl = 0.0;
  for (auto& t : g_timed[slot]) {
    GD_CHECK(hipEventSynchronize(t.b));
    float ms = 0.f;
    GD_CHECK(hipEventElapsedTime(&ms, t.a, t.b));
    tot += ms;
    by += t.bytes;
    fl += t.flops;
  }
  *total_ms = tot;
  *calls = (long long)g_timed[slot].size();
  if (bytes) *bytes = by;
  if (flops) *flops = fl;
  return 0;
}
// summed side-stream bytes of a slot (what its launches move besides the operand / result rows and weights counted in `bytes`)
extern "C" int gdmae_kernel_timing_read_side(int slot, double* side_bytes) {
  GD_REQUIRE(slot >= 0 && slot < GD_T_SLOTS && side_bytes != nullptr, "kernel timing: no such slot");
  double sd = 0.0;
  for (auto& t : g_timed[slot]) sd += t.side;
  *side_bytes = sd;
  return 0;
}
// round-2 names of the two attention slots
extern "C" int gdmae_attention_timing(int on) { return gdmae_kernel_timing(on); }
extern "C" int gdmae_attention_timing_read(int which, double* total_ms, long long* calls) {
  GD_REQUIRE(which == 0 || which == 1, "attention timing: which = 0 (forward) or 1 (backward)");
  return gdmae_kernel_timing_read(which == 0 ? GD_T_ATTN_FWD : GD_T_ATTN_BWD, total_ms, calls, nullptr, nullptr);
}

