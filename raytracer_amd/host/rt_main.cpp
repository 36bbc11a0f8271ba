// Headless counterpart of the reference's main() (src/main.cc:210-239): build the default
// scene with the reference's own construction calls, trace it on the GPU, write the frame
// to a file and print the elapsed time.
//   rt_main [--width W] [--height H] [--depth D] [--device N] [--out frame.bmp|frame.ppm]
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "scene_model.hpp"

// Same objects, lights and literals as the reference's SetupScene (src/main.cc:153-206):
// a grey floor, a glassy violet sphere, an opaque orange sphere behind it, two lights.
static void SetupScene(std::vector<std::unique_ptr<Object>>& objs, std::vector<std::unique_ptr<PointLight>>& lights,
                       Color& bgr) {
    const Light ambient{bgr, 0.3f};
    const Color grey = Color{1.0, 1.0, 1.0} * 0.5;
    const Color violet{0.5, 0.313, 0.64};
    const Color orange{0.9, 0.4, 0.298};

    objs.push_back(std::make_unique<Plane>(gmtl::Vec3d{0, -3, 0}, gmtl::Vec3d{0, 1, 0},
                                           DiffuseMaterial{grey, grey, Color{0, 0, 0} * 0.5, 0, 0}, ambient));
    objs.push_back(std::make_unique<Sphere>(gmtl::Vec3d{2, 0.5, 10}, 1,
                                            DiffuseMaterial{violet, violet, violet, .7, .9f, 1.1}, ambient));
    objs.push_back(std::make_unique<Sphere>(gmtl::Vec3d{1.5, 0, 13}, 1,
                                            DiffuseMaterial{orange, orange, orange, 0.2f, 0.f}, ambient));

    lights.push_back(std::make_unique<PointLight>(Color{1, 1, 1}, 1.f, gmtl::Vec3d{0, 4, 7}));
    lights.push_back(std::make_unique<PointLight>(Color{1, 1, 1}, .3f, gmtl::Vec3d{10, 8, 3}));
}

int main(int argc, char** argv) {
    std::size_t width = 640, height = 640;  // src/main.cc:215
    TracerOptions opt;
    std::string out = "frame.bmp";
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
        else { std::cerr << "usage: rt_main [--width W] [--height H] [--depth D] [--device N] [--out file]\n"; return 2; }
    }

    const auto t0 = std::chrono::high_resolution_clock::now();
    auto screen = std::make_unique<Screen>(width, height, out);

    std::vector<std::unique_ptr<Object>> objs;
    std::vector<std::unique_ptr<PointLight>> lights;
    Color bgr = Color{0.58, 0.76, 0.92};
    screen->ClearScreenColor(bgr);
    SetupScene(objs, lights, bgr);

    try {
        Tracer(bgr, *screen, objs, lights, opt);
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    screen->DrawScreen();

    const auto t1 = std::chrono::high_resolution_clock::now();
    std::cout << "Time: " << std::chrono::duration<double, std::milli>(t1 - t0).count() / 1000.0 << " Seconds\n";
    return 0;
}
