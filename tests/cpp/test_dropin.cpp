// tests/cpp/test_dropin.cpp -- the reference's main.cpp flow (Camera -> RayTracer -> Octree -> loadScene -> setScene -> start ->
// run -> getImage) compiled against the drop-in headers.  argv[1] = .scn, argv[2] = "gpu" to actually render (GPU box),
// otherwise only the host side (loader, Octree::rebuild, API surface, copyability) is exercised.
#include <cstdio>
#include <cstring>
#include <string>
#include <algorithm>
#include <chrono>
#include <thread>
#include "../../include/gi/builtin_loaders.h"

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    Camera camera(gi::dvec3(10, 5, 0), gi::dvec3(0, 0, 0));        // main.cpp:28
    RayTracer raytracer(camera);                                   // main.cpp:30
    Octree* scene = new Octree();                                  // main.cpp:34
    loadScene(scene, raytracer, argv[1]);                          // main.cpp:36-39
    if (argc > 2 && strncmp(argv[2], "gpu", 3) == 0) raytracer.ambient = gi::dvec3(0.05, 0.06, 0.07);   // a sky that is not black: every rendered row shows in the image
    raytracer.setScene(scene);                                     // main.cpp:41
    RayTracer copy = raytracer;                                    // gui.h:19 / viewer.h:16 pass it by value
    scene->rebuild();
    printf("lights %zu photons %d samples %d..%d up %.6f %.6f %.6f\n", scene->lights.size(), copy.photons, copy.min_samples, copy.max_samples,
           copy._camera.up.x, copy._camera.up.y, copy._camera.up.z);
    if (scene->lights.size()) printf("light0 angle %.12f\n", scene->lights[0]->angle);
    if (scene->lights.size()) {   // Light::getPoint() / getPoint(i): the table of 250 directions (include/light.h:17-40)
        Light& l = *scene->lights[0];
        const gi::dvec3 a = l.points[125], g0 = l.getPoint() - l.pos, g1 = l.getPoint(7) - l.pos;
        printf("light0 points %zu p125 %.17g %.17g %.17g getPoint %.12f %.12f rad %.12f\n", l.points.size(), a.x, a.y, a.z,
               std::sqrt(g0.x * g0.x + g0.y * g0.y + g0.z * g0.z), std::sqrt(g1.x * g1.x + g1.y * g1.y + g1.z * g1.z), l.rad);
    }
    // programmatic scene as the mesh generators build it (entities.h:721-738): two triangles and a light
    Octree* quad = new Octree();
    texture white(gi::dvec3(1, 1, 1)), black(gi::dvec3(0, 0, 0));
    Material mat(&white, &black, 1, 1);
    quad->push_back(new triangle(vertex(gi::dvec3(0, 0, 0)), vertex(gi::dvec3(1, 0, 0)), vertex(gi::dvec3(0, 0, 1)), mat));
    quad->push_back(new triangle(vertex(gi::dvec3(1, 0, 0)), vertex(gi::dvec3(1, 0, 1)), vertex(gi::dvec3(0, 0, 1)), mat));
    quad->push_back(new sphere(gi::dvec3(0.5, 0.3, 0.5), 0.3, mat));
    // textured materials as the scene loader builds them (sceneLoader.cpp:47-64): checkerboard and image file
    checkerboard cb(4, gi::dvec3(1, 1, 1), gi::dvec3(0, 0, 1));
    Material cbm(&cb, &black, 1, 1);
    quad->push_back(new triangle(vertex(gi::dvec3(0, 1, 0), gi::dvec3(0, 1, 0), gi::dvec2(0, 0)), vertex(gi::dvec3(1, 1, 0), gi::dvec3(0, 1, 0), gi::dvec2(1, 0)),
                                 vertex(gi::dvec3(0, 1, 1), gi::dvec3(0, 1, 0), gi::dvec2(0, 1)), cbm));
    if (argc > 3) {
        imageTexture* im = new imageTexture(argv[3], gi::dvec2(2, 1));
        gi::dvec2 uv(0.3, 0.4);
        gi::dvec3 px = im->get(uv);
        printf("image %dx%d alpha %d get(0.3,0.4) %.12f %.12f %.12f a %.6f\n", im->width, im->height, (int)im->has_alpha, px.x, px.y, px.z, im->getAlpha(uv));
        Material imm(im, &black, 1, 1);
        quad->push_back(new sphere(gi::dvec3(0.2, 0.3, 0.2), 0.1, imm));
    }
    quad->push_back(new HeightFog(gi::dvec3(0.5, 0.5, 0.5), gi::dvec3(1, 1, 1), gi::dvec3(1, 1, 1), 2, .5, 2));
    quad->push_back(new Light(gi::dvec3(0, 5, 0), gi::dvec3(0, 0, 0), gi::dvec3(4, 4, 4), .05));
    quad->rebuild();
    printf("quad valid %d\n", (int)quad->valid);
    if (argc > 2 && strncmp(argv[2], "gpu", 3) == 0) {
        if (strcmp(argv[2], "gpu2") == 0) copy.devices = {0, 0, 0};   // three contexts on device 0: the several-GPU path of run() on a one-GPU box
        copy.min_samples = copy.max_samples = 4;
        copy.photons = 2000;
        copy.start();
        copy.run(64, 36);                                          // viewer.h:52-56
        if (!copy.last_error().empty()) { printf("error: %s\n", copy.last_error().c_str()); return 1; }
        auto img = copy.getImage();
        double sum = 0;
        for (int y = 0; y < 36; y++) for (int x = 0; x < 64; x++) { gi::dvec3 px = img->getPixel(x, y); sum += px.x + px.y + px.z; }
        double lin = 0;
        for (float v : copy.linear()) lin += v;
        printf("rendered 64x36: 8-bit sum %.6f linear mean %.9f\n", sum, lin / copy.linear().size());
        if (argc > 4) printf("saved %d %d\n", (int)img->save_ppm((std::string(argv[4]) + ".ppm").c_str()), (int)gi_save_pfm((std::string(argv[4]) + ".pfm").c_str(), copy.linear().data(), 64, 36));
        // a second frame after a scene edit: the scene is rebuilt and uploaded again, the photon map is kept (the reference keeps a valid map
        // across frames and never emits twice, include/raytracer.h:61-72)
        const int photons_before = copy.photons_on_device;
        scene->push_back(new triangle(vertex(gi::dvec3(0, 2, 0)), vertex(gi::dvec3(.1, 2, 0)), vertex(gi::dvec3(0, 2, .1)), mat));
        copy.run(64, 36);
        printf("photons before %d after %d valid %d\n", photons_before, copy.photons_on_device, (int)scene->valid);
        // ---- progressive display + cancellation (viewer.h:18-21,36-39, raytracer.h:93-160): the worker fills the shared image stripe by stripe,
        // the "GUI" thread watches it and then stops it
        RayTracer live = raytracer;
        live.devices = copy.devices;
        live.min_samples = live.max_samples = 64;
        live.photons = 2000;
        live.progressive_rows = 16;
        const int W = 512, H = 512;
        live.start();
        std::thread worker([&]() { live.run(W, H); });
        auto filled_rows = [&]() {   // rows of the shared image that hold a non-black pixel
            auto im = live.getImage();
            int n = 0;
            if (im->width() != W) return 0;
            for (int y = 0; y < H; y++) { bool any = false; for (int x = 0; x < W && !any; x++) { gi::dvec3 c = im->getPixel(x, y); any = c.x + c.y + c.z > 0; } n += any; }
            return n;
        };
        int last = 0, polls = 0, monotone = 1, partial_seen = 0;
        while (live.rows_done < 96 && polls < 100000) {
            const int done = live.rows_done, f = filled_rows();
            if (f < last) monotone = 0;
            if (f > 0 && f < H) partial_seen = 1;
            if (f > done + std::max(16, (int)live.rows_in_flight) * (int)std::max<size_t>(1, live.devices.size())) monotone = 0;   // nothing beyond the step in flight (one stripe per device) is ever painted
            last = f; polls++;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        const auto t0 = std::chrono::steady_clock::now();
        live.stop();
        worker.join();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const int done = live.rows_done, f = filled_rows();
        if (f > 0 && f < H) partial_seen = 1;                      // the frame the GUI is left with is the part rendered so far
        printf("progressive: monotone %d partial_seen %d stopped at row %d of %d (filled %d) in %.1f ms running %d\n", monotone, partial_seen, done, H, f, ms, (int)live.running());
        // ---- what the growing stripes are for: a whole frame, progressively, in about the time of one call
        RayTracer whole = raytracer;
        whole.devices = copy.devices;
        whole.min_samples = 8; whole.max_samples = 32;   // the reference's default adaptive setting
        whole.photons = 2000;
        whole.start();
        whole.run(1920, 1080);   // untimed: scene upload, photon emission, the pool's first allocation
        for (int fixed = 0; fixed < 2; fixed++) {
            whole.progressive_ms = fixed ? 0 : 50;
            whole.start();
            const auto t1 = std::chrono::steady_clock::now();
            whole.run(1920, 1080);
            printf("frame 1920x1080 %s stripes: %.0f ms\n", fixed ? "16-row" : "growing", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        }
    }
    return 0;
}
