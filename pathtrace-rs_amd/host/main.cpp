// main.cpp -- CLI with the reference's flags (main.rs:26-96). The preview window
// (glium_window.rs) is out of scope: without -O the tool still renders offline,
// and `-F frames` accumulates that many progressive frames before saving.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "host.hpp"

static void usage() {
    puts("Toy Path Tracer 0.1 (MI355X / HIP back end)\n"
         "USAGE: pathtrace [FLAGS] [OPTIONS]\n"
         "  -W, --width <width>      Image width to generate [1280]\n"
         "  -H, --height <height>    Image height to generate [720]\n"
         "  -S, --samples <samples>  Number of samples per pixel [4]\n"
         "  -D, --depth <depth>      Maximum bounces per ray [10]\n"
         "  -R, --random             Use a random seed\n"
         "  -P, --preset <preset>    Scene preset to render [two_perlin_spheres]\n"
         "  -F, --frames <frames>    Process a fixed number of frames and exit\n"
         "  -B, --bvh                Use bounding volume hierarchy instead of a flat list\n"
         "  -O, --offline            Don't create a preview render window (always the case here)\n"
         "      --device <n>         HIP device ordinal [0]\n"
         "      --output <path>      PNG path [output.png]");
}

int main(int argc, char **argv) {
    pt::Params params;  // defaults main.rs:78-85
    std::string preset = "two_perlin_spheres", output = "output.png";
    uint32_t frames = 1;
    int device = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](uint32_t &dst) {
            if (i + 1 >= argc) { fprintf(stderr, "error: %s requires a value\n", a.c_str()); exit(2); }
            dst = (uint32_t)strtoul(argv[++i], nullptr, 10);
        };
        if (a == "-W" || a == "--width") value(params.width);
        else if (a == "-H" || a == "--height") value(params.height);
        else if (a == "-S" || a == "--samples") value(params.samples);
        else if (a == "-D" || a == "--depth") value(params.max_depth);
        else if (a == "-F" || a == "--frames") value(frames);
        else if (a == "-R" || a == "--random") params.random_seed = true;
        else if (a == "-B" || a == "--bvh") params.use_bvh = true;
        else if (a == "-O" || a == "--offline") {}
        else if (a == "-X" || a == "--print") { fprintf(stderr, "-X (BVH debug trace) is not part of the accelerated path\n"); return 2; }
        else if ((a == "-P" || a == "--preset") && i + 1 < argc) preset = argv[++i];
        else if (a == "--output" && i + 1 < argc) output = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); usage(); return 2; }
    }
    return pt::render_offline(preset, params, device, output, frames);
}
