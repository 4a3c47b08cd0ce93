// Drives the C++ host mirror (include/fic_host.hpp) the way RLEAppController.openDecodedImage
// (RLEAppController.java:172-188) drives the reference: set the two statics, encode(image, out), decode(in).
//   host_mirror_test encode      <gray.raw u8>   w h B wK out.run collage.raw
//   host_mirror_test encode_argb <argb.raw i32>  w h B wK out.run collage.raw     (colour -> encodeRGB)
//   host_mirror_test encode_multi <gray.raw u8>  w h B wK n_iso n_gpus out.run     (fic_encode_gray_argb_multi: what the
//                                                                                  JNI host calls on a multi-GPU node)
//   host_mirror_test decode      <in.run> out_argb.raw                            (prints avgError)
//   host_mirror_test synth       U|S w h seed out.raw                             (include/fic_synth.h; no GPU needed)
//   host_mirror_test kernel      Dw Dh index wK                                   (prints generateKernel's dy dx; no GPU needed)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include "fic_host.hpp"
#include "fic_synth.h"

using FC = bvk_ss19::FractalCompression;

static int run(int argc, char** argv)
{
    if (argc == 6 && !std::strcmp(argv[1], "kernel")) {
        FC::widthKernel = std::atoi(argv[5]);
        auto k = FC::generateKernel(std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]));
        std::printf("%d %d\n", k[0], k[1]);
        return 0;
    }
    if (argc == 7 && !std::strcmp(argv[1], "synth")) {
        const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
        std::vector<uint8_t> img((size_t)w * h);
        fic_synth_image(argv[2][0], w, h, std::strtoull(argv[5], nullptr, 0), img.data());
        std::ofstream o(argv[6], std::ios::binary);
        o.write(reinterpret_cast<const char*>(img.data()), (std::streamsize)img.size());
        return 0;
    }
    if (argc == 10 && !std::strcmp(argv[1], "encode_multi")) {
        const int w = std::atoi(argv[3]), h = std::atoi(argv[4]), B = std::atoi(argv[5]), wK = std::atoi(argv[6]);
        const int n_iso = std::atoi(argv[7]), n_gpus = std::atoi(argv[8]);
        std::vector<unsigned char> g((size_t)w * h);
        std::ifstream in(argv[2], std::ios::binary);
        in.read(reinterpret_cast<char*>(g.data()), (std::streamsize)g.size());
        if (!in) { std::fprintf(stderr, "short read\n"); return 2; }
        std::vector<int32_t> argb(g.size());
        for (size_t i = 0; i < g.size(); i++)
            argb[i] = (int32_t)(0xff000000u | ((uint32_t)g[i] << 16) | ((uint32_t)g[i] << 8) | g[i]);
        int Rw = 0, Rh = 0;
        if (fic_geometry(w, h, B, &Rw, &Rh, nullptr, nullptr) < 0) throw std::runtime_error(fic_last_error());
        const int nr = Rw * Rh;
        std::vector<int32_t> idx(nr), iso(nr), q((size_t)nr * 3);
        std::vector<float> a(nr), b(nr);
        if (fic_encode_gray_argb_multi(argb.data(), w, h, B, wK, n_iso, n_gpus, idx.data(), a.data(), b.data(), iso.data(), q.data()) < 0)
            throw std::runtime_error(fic_last_error());
        std::vector<uint8_t> buf(20 + 12 * (size_t)nr);
        const int64_t n = fic_write_run_gray(q.data(), nr, w, h, B, wK, buf.data(), (int64_t)buf.size());
        if (n < 0) throw std::runtime_error(fic_last_error());
        std::ofstream o(argv[9], std::ios::binary);
        o.write(reinterpret_cast<const char*>(buf.data()), n);
        o.write(reinterpret_cast<const char*>(iso.data()), (std::streamsize)(iso.size() * 4));     // + isometry ids, for the test
        o.write(reinterpret_cast<const char*>(a.data()), (std::streamsize)(a.size() * 4));
        o.write(reinterpret_cast<const char*>(b.data()), (std::streamsize)(b.size() * 4));
        return 0;
    }
    if (argc >= 4 && !std::strcmp(argv[1], "decode")) {
        std::ifstream in(argv[2], std::ios::binary);
        bvk_ss19::RasterImage img = FC::decode(in);
        std::ofstream o(argv[3], std::ios::binary);
        o.write(reinterpret_cast<const char*>(img.argb.data()), (std::streamsize)(img.argb.size() * 4));
        std::printf("%d %d %.9g\n", img.width, img.height, (double)FC::getAvgError());
        return 0;
    }
    if (argc != 9 || (std::strcmp(argv[1], "encode") && std::strcmp(argv[1], "encode_argb"))) {
        std::fprintf(stderr, "usage: %s encode|encode_argb in.raw w h B wK out.run collage.raw | decode in.run out.raw\n", argv[0]);
        return 2;
    }
    const bool argb = !std::strcmp(argv[1], "encode_argb");
    int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
    bvk_ss19::RasterImage img(w, h);
    std::ifstream in(argv[2], std::ios::binary);
    if (argb) {
        in.read(reinterpret_cast<char*>(img.argb.data()), (std::streamsize)(img.argb.size() * 4));
    } else {
        std::vector<unsigned char> g((size_t)w * h);
        in.read(reinterpret_cast<char*>(g.data()), (std::streamsize)g.size());
        for (size_t i = 0; i < g.size(); i++)
            img.argb[i] = (int32_t)(0xff000000u | ((uint32_t)g[i] << 16) | ((uint32_t)g[i] << 8) | g[i]);
    }
    if (!in) { std::fprintf(stderr, "short read\n"); return 2; }
    FC::blockgroesse = std::atoi(argv[5]);
    FC::widthKernel = std::atoi(argv[6]);
    std::ofstream out(argv[7], std::ios::binary);
    bvk_ss19::RasterImage collage = FC::encode(img, out);
    std::ofstream c(argv[8], std::ios::binary);
    c.write(reinterpret_cast<const char*>(collage.argb.data()), (std::streamsize)(collage.argb.size() * 4));
    return 0;
}

int main(int argc, char** argv)
{
    try {
        return run(argc, argv);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
}
