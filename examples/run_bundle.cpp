// run_bundle.cpp -- FastDepth inference from a deploy bundle with no Python and no nn.Module: the MI355X counterpart of the reference's
// TX2 runner (deploy/tx2_run_tvm.py: load graph + params, feed one frame, save the output, warm up, time repeated runs).
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/run_bundle.cpp -L fast-depth_amd/fastdepth_hip -lfastdepth_hip -o run_bundle
//   ./run_bundle model.fdplan input_f32_nchw.bin output_f32.bin [warmup_trials] [run_trials]
//
// input: raw float32 [B,3,H,W] (values in [0,1]) for the bundle's planned shape; output: raw float32 [B,1,H,W].
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fastdepth_hip.h"

#define CHECK_FD(x) do { int rc_ = (x); if (rc_ != FD_OK) { fprintf(stderr, "%s failed: %s\n", #x, fd_last_error()); return 2; } } while (0)
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

static bool read_file(const char *path, std::vector<unsigned char> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    out.resize(n);
    const bool ok = fread(out.data(), 1, n, f) == (size_t)n;
    fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s bundle.fdplan input.bin output.bin [warmup_trials=10] [run_trials=100]\n", argv[0]); return 1; }
    const int warmup = argc > 4 ? atoi(argv[4]) : 10, trials = argc > 5 ? atoi(argv[5]) : 100;
    std::vector<unsigned char> bundle, input;
    if (!read_file(argv[1], bundle) || !read_file(argv[2], input)) { fprintf(stderr, "cannot read the bundle / input file\n"); return 1; }
    printf("=> [HIP on MI355X] %s, bundle of %zu bytes\n", fd_version(), bundle.size());
    fd_plan *plan = nullptr;
    CHECK_FD(fd_plan_import(bundle.data(), bundle.size(), 0, &plan));
    int32_t b, h, w, dt;
    CHECK_FD(fd_plan_shape(plan, &b, &h, &w, &dt));
    const size_t in_bytes = (size_t)b * 3 * h * w * 4, out_bytes = (size_t)b * h * w * 4;
    if (input.size() != in_bytes) { fprintf(stderr, "input has %zu bytes, the bundle is planned for [%d,3,%d,%d] float32 = %zu\n", input.size(), b, h, w, in_bytes); return 1; }
    void *ws = nullptr, *x = nullptr, *y = nullptr;
    CHECK_HIP(hipMalloc(&ws, fd_plan_workspace_bytes(plan)));
    CHECK_HIP(hipMalloc(&x, in_bytes)); CHECK_HIP(hipMalloc(&y, out_bytes));
    CHECK_FD(fd_plan_bind_workspace(plan, ws, fd_plan_workspace_bytes(plan)));
    CHECK_FD(fd_plan_import_weights(plan, bundle.data(), bundle.size(), nullptr));
    CHECK_HIP(hipMemcpy(x, input.data(), in_bytes, hipMemcpyHostToDevice));
    CHECK_FD(fd_forward(plan, x, y, nullptr));
    std::vector<unsigned char> out(out_bytes);
    CHECK_HIP(hipMemcpy(out.data(), y, out_bytes, hipMemcpyDeviceToHost));
    FILE *f = fopen(argv[3], "wb");
    if (!f || fwrite(out.data(), 1, out_bytes, f) != out_bytes) { fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
    fclose(f);
    printf("=> [HIP on MI355X] benchmarking: %d warmup, %d run trials\n", warmup, trials);
    for (int i = 0; i < warmup; ++i) { CHECK_FD(fd_forward(plan, x, y, nullptr)); CHECK_HIP(hipDeviceSynchronize()); }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < trials; ++i) CHECK_FD(fd_forward(plan, x, y, nullptr));
    CHECK_HIP(hipDeviceSynchronize());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / trials;
    printf("=> [HIP on MI355X] profiled runtime (in ms): %.5f  (batch %d -> %.1f frames/s)\n", ms, b, b / ms * 1e3);
    fd_plan_destroy(plan);
    (void)hipFree(ws); (void)hipFree(x); (void)hipFree(y);
    return 0;
}
