// Drives the C++ host mirror (include/fic_host.hpp) the way RLEAppController.openDecodedImage
// (RLEAppController.java:172-188) drives the reference: set the two statics, call encode(image, out).
// Usage: host_mirror_test <gray.raw> <w> <h> <B> <wK> <out.run> <collage.raw>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include "fic_host.hpp"

int main(int argc, char** argv)
{
    if (argc != 8) { std::fprintf(stderr, "usage: %s gray.raw w h B wK out.run collage.raw\n", argv[0]); return 2; }
    int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    std::vector<unsigned char> g((size_t)w * h);
    std::ifstream in(argv[1], std::ios::binary);
    in.read(reinterpret_cast<char*>(g.data()), (std::streamsize)g.size());
    if (!in) { std::fprintf(stderr, "short read\n"); return 2; }
    bvk_ss19::RasterImage img(w, h);
    for (size_t i = 0; i < g.size(); i++)
        img.argb[i] = (int32_t)(0xff000000u | ((uint32_t)g[i] << 16) | ((uint32_t)g[i] << 8) | g[i]);
    using FC = bvk_ss19::FractalCompression;
    FC::blockgroesse = std::atoi(argv[4]);
    FC::widthKernel = std::atoi(argv[5]);
    try {
        std::ofstream out(argv[6], std::ios::binary);
        bvk_ss19::RasterImage collage = FC::encode(img, out);
        std::ofstream c(argv[7], std::ios::binary);
        c.write(reinterpret_cast<const char*>(collage.argb.data()), (std::streamsize)(collage.argb.size() * 4));
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
