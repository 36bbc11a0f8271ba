// Headless counterpart of the reference's main() (src/main.cc:210-239): build a scene with
// the reference's own construction calls, trace it on the GPU, write the frame to a file
// and print the elapsed time.
//   rt_main [--scene default|spheres1024|glass] [--orbit DEGREES] [--width W] [--height H]
//           [--depth D] [--device N] [--frames N] [--out frame.bmp|frame.ppm]
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "scene_builders.hpp"
#include "scene_model.hpp"

int main(int argc, char** argv) {
    std::size_t width = 640, height = 640;  // src/main.cc:215
    TracerOptions opt;
    std::string out = "frame.bmp";
    std::string scene = "default";
    double orbit_deg = 0.0;
    int frames = 1;
    for (int i = 1; i < argc; ++i) {
        auto next = [&]() -> const char* {
            if (i + 1 >= argc) { std::cerr << "missing value after " << argv[i] << "\n"; std::exit(2); }
            return argv[++i];
        };
        if (!std::strcmp(argv[i], "--width")) width = std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--height")) height = std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--depth")) opt.max_depth = std::atoi(next());
        else if (!std::strcmp(argv[i], "--device")) opt.device = std::atoi(next());
        else if (!std::strcmp(argv[i], "--out")) out = next();
        else if (!std::strcmp(argv[i], "--scene")) scene = next();
        else if (!std::strcmp(argv[i], "--orbit")) orbit_deg = std::atof(next());
        else if (!std::strcmp(argv[i], "--frames")) frames = std::atoi(next());
        else {
            std::cerr << "usage: rt_main [--scene default|spheres1024|glass] [--orbit DEGREES] [--width W] "
                         "[--height H] [--depth D] [--device N] [--frames N] [--out file]\n";
            return 2;
        }
    }

    const auto t0 = std::chrono::high_resolution_clock::now();
    auto screen = std::make_unique<Screen>(width, height, out);

    std::vector<std::unique_ptr<Object>> objs;
    std::vector<std::unique_ptr<PointLight>> lights;
    Color bgr = Color{0.58, 0.76, 0.92};
    screen->ClearScreenColor(bgr);
    if (scene == "default") SetupScene(objs, lights, bgr);
    else if (scene == "spheres1024") SetupRandomSpheresScene(objs, lights, bgr);
    else if (scene == "glass") SetupGlassScene(objs, lights, bgr);
    else { std::cerr << "unknown scene " << scene << "\n"; return 2; }
    if (orbit_deg != 0.0) OrbitScene(objs, lights, orbit_deg * 3.141592653589793 / 180.0);

    const auto setup_s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    const auto tr0 = std::chrono::high_resolution_clock::now();
    TracerTimings first, next;
    double first_s = 0.0, next_s = 0.0;
    try {
        opt.timings = &first;
        Tracer(bgr, *screen, objs, lights, opt);
        first_s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - tr0).count();
        opt.timings = &next;
        for (int f = 1; f < frames; ++f) {  // more frames of the same scene: what a sequence pays per frame
            screen->DrawScreen();
            const auto tn = std::chrono::high_resolution_clock::now();
            Tracer(bgr, *screen, objs, lights, opt);
            next_s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - tn).count();
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    const auto tr1 = std::chrono::high_resolution_clock::now();
    screen->DrawScreen();

    const auto t1 = std::chrono::high_resolution_clock::now();
    std::cout << "Setup: " << setup_s << " Seconds (screen buffers, scene construction)\n";
    std::cout << "TracerFirst: " << first_s << " Seconds (rt_create " << first.create_s << ", rt_upload_scene "
              << first.upload_s << ", rt_render " << first.render_s << ")\n";
    if (frames > 1)
        std::cout << "TracerNext: " << next_s << " Seconds (rt_render " << next.render_s << "; the last of " << frames - 1
                  << " further call(s))\n";
    std::cout << "Tracer: " << std::chrono::duration<double, std::milli>(tr1 - tr0).count() / 1000.0 << " Seconds ("
              << frames << " call(s), first one with device set-up)\n";
    std::cout << "Time: " << std::chrono::duration<double, std::milli>(t1 - t0).count() / 1000.0 << " Seconds\n";
    ReleaseTracerDevices();
    return 0;
}
