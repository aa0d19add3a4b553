// Socket power / shader clock / power-limit residency of device 0 while a harness runs its timed loop (make
// "power-limited" a measurement).  Not part of the library.  librocm_smi64 is bound with dlopen (the harnesses stay buildable on a
// box without it); the sampler thread polls rsmi_dev_gpu_metrics_info_get every few milliseconds between start() and stop():
//   current_socket_power [W], current_gfxclks[0..7] [MHz] (one per XCD), and the firmware's accumulators -- energy_accumulator
//   (15.259 uJ units), ppt_residency_acc / socket_thm_residency_acc / prochot_residency_acc (time the power / thermal limiters were
//   active, in accumulation_counter ticks) -- whose DELTAS over the window say whether the package power tracker was what held the
//   clock.  The cap is rsmi_dev_power_cap_get.
#pragma once
#include <dlfcn.h>
#include <stdint.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <rocm_smi/rocm_smi.h>

struct SmiSampler {
  void *h = nullptr;
  rsmi_status_t (*p_init)(uint64_t) = nullptr;
  rsmi_status_t (*p_metrics)(uint32_t, rsmi_gpu_metrics_t *) = nullptr;
  rsmi_status_t (*p_cap)(uint32_t, uint32_t, uint64_t *) = nullptr;
  rsmi_status_t (*p_power)(uint32_t, uint64_t *, RSMI_POWER_TYPE *) = nullptr;
  bool ok = false;
  std::atomic<bool> run{false};
  std::thread th;
  std::vector<double> pw, clk;
  rsmi_gpu_metrics_t first, last;
  bool have_first = false;
  uint64_t cap_uw = 0;
  int period_ms = 5;

  bool open() {
    const char *names[] = {"librocm_smi64.so.1", "librocm_smi64.so", "/opt/rocm/lib/librocm_smi64.so"};
    for (const char *nm : names) {
      h = dlopen(nm, RTLD_NOW);
      if (h) break;
    }
    if (!h) return false;
    p_init = reinterpret_cast<decltype(p_init)>(dlsym(h, "rsmi_init"));
    p_metrics = reinterpret_cast<decltype(p_metrics)>(dlsym(h, "rsmi_dev_gpu_metrics_info_get"));
    p_cap = reinterpret_cast<decltype(p_cap)>(dlsym(h, "rsmi_dev_power_cap_get"));
    p_power = reinterpret_cast<decltype(p_power)>(dlsym(h, "rsmi_dev_power_get"));
    if (!p_init || !p_metrics) return false;
    if (p_init(0) != RSMI_STATUS_SUCCESS) return false;
    rsmi_gpu_metrics_t m;
    memset(&m, 0, sizeof m);
    if (p_metrics(0, &m) != RSMI_STATUS_SUCCESS) return false;
    if (p_cap) p_cap(0, 0, &cap_uw);
    ok = true;
    return true;
  }
  static double mean_clk(const rsmi_gpu_metrics_t &m) {
    double s = 0;
    int k = 0;
    for (int i = 0; i < 8; ++i)
      if (m.current_gfxclks[i] != 0 && m.current_gfxclks[i] != 0xFFFF) { s += m.current_gfxclks[i]; ++k; }
    if (k) return s / k;
    return m.current_gfxclk != 0xFFFF ? m.current_gfxclk : 0.0;
  }
  void start() {
    if (!ok) return;
    pw.clear();
    clk.clear();
    have_first = false;
    run = true;
    th = std::thread([this] {
      while (run.load()) {
        rsmi_gpu_metrics_t m;
        memset(&m, 0, sizeof m);
        if (p_metrics(0, &m) == RSMI_STATUS_SUCCESS) {
          if (!have_first) { first = m; have_first = true; }
          last = m;
          double w = m.current_socket_power != 0xFFFF ? m.current_socket_power : m.average_socket_power;
          pw.push_back(w);
          clk.push_back(mean_clk(m));
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(period_ms));
      }
    });
  }
  // prints one line; `skip` = fraction of the window's first samples left out of the averages (ramp)
  void stop(const char *tag, double skip = 0.25) {
    if (!ok) {
      printf("[smi %s] unavailable (librocm_smi64 / gpu_metrics not readable on this box)\n", tag);
      return;
    }
    run = false;
    th.join();
    const size_t n = pw.size(), s0 = (size_t)(n * skip);
    double ps = 0, pm = 0, cs = 0;
    size_t k = 0;
    for (size_t i = s0; i < n; ++i) { ps += pw[i]; cs += clk[i]; if (pw[i] > pm) pm = pw[i]; ++k; }
    if (!k) { printf("[smi %s] no samples\n", tag); return; }
    const double ticks = (double)(last.accumulation_counter - first.accumulation_counter);
    const double ppt = (double)(last.ppt_residency_acc - first.ppt_residency_acc);
    const double thm = (double)(last.socket_thm_residency_acc - first.socket_thm_residency_acc);
    const double hot = (double)(last.prochot_residency_acc - first.prochot_residency_acc);
    const double dt_s = (double)(last.firmware_timestamp - first.firmware_timestamp) * 1e-8;  // 10 ns units
    const double de_j = (double)(last.energy_accumulator - first.energy_accumulator) * 15.259e-6;
    printf("[smi %s] samples %zu  socket power avg %.0f W max %.0f W  cap %.0f W  gfxclk avg %.0f MHz  | firmware window %.3f s: energy %.1f J "
           "(%.0f W)  ppt-limited %.3f  thermal-limited %.3f  prochot %.3f of %.0f ticks  throttle_status 0x%llx\n",
           tag, k, ps / k, pm, cap_uw * 1e-6, cs / k, dt_s, de_j, dt_s > 0 ? de_j / dt_s : 0.0, ticks > 0 ? ppt / ticks : 0.0,
           ticks > 0 ? thm / ticks : 0.0, ticks > 0 ? hot / ticks : 0.0, ticks, (unsigned long long)last.indep_throttle_status);
  }
};
