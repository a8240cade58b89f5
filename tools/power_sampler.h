// Energy per launch for the stand-alone harnesses (round 6: power as a bound -- the training step draws ~1.35 kW at ~1.9 GHz, so a change that moves
// fewer bytes into LDS per MFMA can win on clock what it ties on cycles).  A host thread samples the amdgpu hwmon files of the device the harness runs
// on (power1_input: socket power in microwatts; freq1_input: shader clock in Hz; /sys/class/drm/card*/device/hwmon/hwmon*) every 5 ms while `f` is
// launched back to back for `seconds`; energy per launch = mean power x mean launch time.  Returns false when the files are not readable.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <functional>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>
#include <hip/hip_runtime.h>

struct PowerResult { double watts = 0, mhz = 0, us_per_launch = 0, joules_per_launch = 0; int samples = 0; bool ok = false; };

static inline std::string hwmon_dir_of_current_device() {
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return "";
    char bus[64] = {0}; if (hipDeviceGetPCIBusId(bus, sizeof bus, dev) != hipSuccess) return "";
    std::string want(bus); for (auto& ch : want) ch = (char)tolower((unsigned char)ch);
    glob_t g; std::string found;
    if (glob("/sys/class/drm/card*/device/hwmon/hwmon*", 0, nullptr, &g) == 0) {
        for (size_t i = 0; i < g.gl_pathc; ++i) {
            std::string p = g.gl_pathv[i];
            char real[4096]; std::string devlink = p.substr(0, p.find("/hwmon"));
            if (realpath(devlink.c_str(), real)) { std::string r(real); for (auto& ch : r) ch = (char)tolower((unsigned char)ch); if (r.find(want) != std::string::npos) { found = p; break; } }
        }
        if (found.empty() && g.gl_pathc > 0) found = g.gl_pathv[0];
        globfree(&g);
    }
    return found;
}
static inline double read_number(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r"); if (!f) return -1.0;
    double v = -1.0; if (fscanf(f, "%lf", &v) != 1) v = -1.0; fclose(f); return v;
}
static inline PowerResult measure_power(const std::function<void()>& f, double seconds = 1.5) {
    PowerResult r;
    static const std::string dir = hwmon_dir_of_current_device();
    if (dir.empty() || read_number(dir + "/power1_input") < 0) return r;
    std::atomic<bool> stop{false};
    std::vector<double> pw, fq;
    f(); (void)hipDeviceSynchronize();
    std::thread th([&] {
        std::this_thread::sleep_for(std::chrono::milliseconds(150));      // let power and clock settle under the load
        while (!stop.load()) { pw.push_back(read_number(dir + "/power1_input") * 1e-6); fq.push_back(read_number(dir + "/freq1_input") * 1e-6); std::this_thread::sleep_for(std::chrono::milliseconds(5)); }
    });
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0; (void)hipEventRecord(e0, 0);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) { for (int i = 0; i < 50; ++i) f(); n += 50; (void)hipStreamSynchronize(0); }
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    stop.store(true); th.join();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (pw.empty()) return r;
    for (double v : pw) r.watts += v; r.watts /= pw.size();
    for (double v : fq) r.mhz += v; r.mhz /= fq.size();
    r.samples = (int)pw.size(); r.us_per_launch = ms * 1e3 / n; r.joules_per_launch = r.watts * r.us_per_launch * 1e-6; r.ok = true;
    return r;
}
static inline void report_power(const char* what, const PowerResult& r, double flops = 0.0, double bytes = 0.0) {
    if (!r.ok) { printf("energy %-44s hwmon power files not readable\n", what); return; }
    printf("energy %-44s %7.1f us  %6.0f W  %5.0f MHz  %7.4f J per launch", what, r.us_per_launch, r.watts, r.mhz, r.joules_per_launch);
    if (flops > 0) printf("  %5.2f pJ/FLOP", r.joules_per_launch / flops * 1e12);
    if (bytes > 0) printf("  %5.1f pJ/B", r.joules_per_launch / bytes * 1e12);
    printf("  (%d samples)\n", r.samples);
}
