// One frame split into interleaved bands over the GPUs of a node and gathered with RCCL, from
// C++: the reference's main() flow (src/main.cc:210-239) with Tracer()'s row loop
// (src/main.cc:130) dealt out band b -> GPU b mod N.  One process drives all GPUs
// (ncclCommInitAll); every GPU renders its strip with rt_render_device straight into the buffer
// that ncclGather sends to GPU 0 over xGMI; GPU 0 copies the gathered strips to the host,
// where the bands are put back in frame order.
//   rt_gather [--scene default|spheres1024|glass] [--width W] [--height H] [--depth D]
//             [--gpus N] [--band-rows R] [--frames K] [--out frame.bmp]
// Prints the time per frame (render + gather + copy to the host) and writes the last frame.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "scene_builders.hpp"
#include "scene_model.hpp"

#define HIP_OK(call)                                                                        \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) { std::cerr << #call << ": " << hipGetErrorString(e__) << "\n"; return 1; } \
    } while (0)
#define NCCL_OK(call)                                                                       \
    do {                                                                                    \
        ncclResult_t r__ = (call);                                                          \
        if (r__ != ncclSuccess) { std::cerr << #call << ": " << ncclGetErrorString(r__) << "\n"; return 1; } \
    } while (0)
#define RT_CHECK(ctx, call)                                                                 \
    do {                                                                                    \
        if ((call) != RT_OK) { std::cerr << #call << ": " << rt_last_error(ctx) << "\n"; return 1; } \
    } while (0)

int main(int argc, char** argv) {
    int width = 3840, height = 2160, depth = 5, gpus = 0, band_rows = 8, frames = 3;
    std::string out = "frame.bmp", scene = "default";
    for (int i = 1; i < argc; ++i) {
        auto next = [&]() -> const char* {
            if (i + 1 >= argc) { std::cerr << "missing value after " << argv[i] << "\n"; std::exit(2); }
            return argv[++i];
        };
        if (!std::strcmp(argv[i], "--width")) width = std::atoi(next());
        else if (!std::strcmp(argv[i], "--height")) height = std::atoi(next());
        else if (!std::strcmp(argv[i], "--depth")) depth = std::atoi(next());
        else if (!std::strcmp(argv[i], "--gpus")) gpus = std::atoi(next());
        else if (!std::strcmp(argv[i], "--band-rows")) band_rows = std::atoi(next());
        else if (!std::strcmp(argv[i], "--frames")) frames = std::atoi(next());
        else if (!std::strcmp(argv[i], "--out")) out = next();
        else if (!std::strcmp(argv[i], "--scene")) scene = next();
        else { std::cerr << "usage: see the head of rt_gather.cpp\n"; return 2; }
    }
    int n_dev = 0;
    HIP_OK(hipGetDeviceCount(&n_dev));
    if (gpus <= 0 || gpus > n_dev) gpus = n_dev;
    if (gpus < 1 || band_rows < 1 || frames < 1) { std::cerr << "no GPU / bad argument\n"; return 2; }
    const int N = gpus;

    // the scene, built the reference's way and flattened once
    ObjectList objs;
    LightList lights;
    Color bgr = Color{0.58, 0.76, 0.92};
    if (scene == "default") SetupScene(objs, lights, bgr);
    else if (scene == "spheres1024") SetupRandomSpheresScene(objs, lights, bgr);
    else if (scene == "glass") SetupGlassScene(objs, lights, bgr);
    else { std::cerr << "unknown scene " << scene << "\n"; return 2; }
    std::vector<rt_prim> prims;
    for (const auto& o : objs) prims.push_back(o->Flatten());
    std::vector<rt_light> lts;
    for (const auto& l : lights) lts.push_back(l->Flatten());

    // the partition of rank r: bands r, r + N, ... of `band_rows` rows (the last one clipped)
    const int n_bands = (height + band_rows - 1) / band_rows;
    std::vector<rt_params> part(N);
    size_t max_rows = 0;
    for (int r = 0; r < N; ++r) {
        rt_params p{};
        p.width = width; p.height = height; p.max_depth = depth; p.fov_deg = 60;
        p.band_rows = band_rows; p.band_first = r; p.band_step = N;
        p.band_count = (r < n_bands) ? (n_bands - r + N - 1) / N : 0;
        p.bgr[0] = bgr.r(); p.bgr[1] = bgr.g(); p.bgr[2] = bgr.b();
        part[r] = p;
        max_rows = std::max<size_t>(max_rows, static_cast<size_t>(rt_strip_rows(&p)));
    }
    const size_t strip_bytes = max_rows * static_cast<size_t>(width) * 3;  // what every rank sends

    std::vector<rt_ctx*> ctx(N, nullptr);
    std::vector<hipStream_t> stream(N);
    std::vector<uint8_t*> send(N, nullptr);
    uint8_t* recv = nullptr;  // on GPU 0: N strips one after the other
    std::vector<int> devs(N);
    for (int r = 0; r < N; ++r) {
        devs[r] = r;
        if (rt_create(r, &ctx[r]) != RT_OK) { std::cerr << "rt_create: " << rt_last_error(nullptr) << "\n"; return 1; }
        RT_CHECK(ctx[r], rt_upload_scene(ctx[r], prims.data(), static_cast<int32_t>(prims.size()), lts.data(),
                                         static_cast<int32_t>(lts.size())));
        HIP_OK(hipSetDevice(r));
        HIP_OK(hipStreamCreateWithFlags(&stream[r], hipStreamNonBlocking));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&send[r]), strip_bytes ? strip_bytes : 1));
        HIP_OK(hipMemset(send[r], 0, strip_bytes ? strip_bytes : 1));
        if (r == 0) HIP_OK(hipMalloc(reinterpret_cast<void**>(&recv), (strip_bytes ? strip_bytes : 1) * N));
    }
    std::vector<ncclComm_t> comm(N);
    NCCL_OK(ncclCommInitAll(comm.data(), N, devs.data()));

    void* gathered = nullptr;  // page-locked host copy of what GPU 0 received
    if (rt_host_alloc(strip_bytes * N + 1, &gathered) != RT_OK) { std::cerr << rt_last_error(nullptr) << "\n"; return 1; }
    std::vector<uint8_t> frame(static_cast<size_t>(width) * height * 3);
    double last_ms = 0.0, best_ms = 1e30;
    for (int f = 0; f < frames; ++f) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < N; ++r)
            if (part[r].band_count > 0)
                RT_CHECK(ctx[r], rt_render_device(ctx[r], &part[r], stream[r], send[r], nullptr, nullptr, nullptr, nullptr));
        NCCL_OK(ncclGroupStart());
        for (int r = 0; r < N; ++r)
            NCCL_OK(ncclGather(send[r], r == 0 ? recv : nullptr, strip_bytes, ncclUint8, 0, comm[r], stream[r]));
        NCCL_OK(ncclGroupEnd());
        HIP_OK(hipSetDevice(0));
        HIP_OK(hipMemcpyAsync(gathered, recv, strip_bytes * N, hipMemcpyDeviceToHost, stream[0]));
        for (int r = 0; r < N; ++r) {
            HIP_OK(hipSetDevice(r));
            HIP_OK(hipStreamSynchronize(stream[r]));
        }
        // bands back into frame order: band b is band b / N of rank b mod N's strip
        const size_t band_bytes = static_cast<size_t>(band_rows) * width * 3;
        for (int b = 0; b < n_bands; ++b) {
            const size_t y0 = static_cast<size_t>(b) * band_rows;
            const size_t rows = std::min<size_t>(band_rows, height - y0);
            std::memcpy(frame.data() + y0 * width * 3,
                        static_cast<const uint8_t*>(gathered) + (b % N) * strip_bytes + (b / N) * band_bytes, rows * width * 3);
        }
        last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (f > 0) best_ms = std::min(best_ms, last_ms);
    }
    std::cout << "Gather: " << N << " GPU(s), " << width << "x" << height << " depth " << depth << ", bands of " << band_rows
              << " rows; last frame " << last_ms << " ms, best after the first " << (frames > 1 ? best_ms : last_ms) << " ms\n";

    Screen screen(static_cast<std::size_t>(width), static_cast<std::size_t>(height), out);
    screen.StoreRows(0, static_cast<std::size_t>(height), 1, frame.data());
    screen.DrawScreen();

    for (int r = 0; r < N; ++r) {
        ncclCommDestroy(comm[r]);
        (void)hipSetDevice(r);
        (void)hipFree(send[r]);
        (void)hipStreamDestroy(stream[r]);
        rt_destroy(ctx[r]);
    }
    (void)hipSetDevice(0);
    (void)hipFree(recv);
    rt_host_free(gathered);
    return 0;
}
