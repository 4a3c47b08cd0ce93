// fic_host.hpp -- header-only C++ host mirror of the reference's Java entry point, over the C ABI
// of fic.h.  The reference is compiled (Java) code and this image has no JDK, so this is the
// compiled host side a reader can build and run here; the Java/JNI binding of the same calls is in
// fractal-image-compression_amd/java + jni.
//
// Mirrors, with the same names, argument meaning and error behaviour:
//   bvk_ss19.RasterImage            RasterImage.java:18-24   (argb, width, height)
//   bvk_ss19.FractalCompression     FractalCompression.java  blockgroesse :14, widthKernel :15,
//                                   isGreyScale :32, encode :54, encodeGrayScale :109, writeData :230
// The reference throws unchecked exceptions on bad geometry; here every negative C-ABI code
// becomes a thrown std::runtime_error carrying fic_last_error().
#pragma once
#include <cstdint>
#include <istream>
#include <iterator>
#include <ostream>
#include <stdexcept>
#include <string>
#include <array>
#include <vector>

#include "fic.h"

namespace bvk_ss19 {

struct RasterImage {            // RasterImage.java:18-24
    std::vector<int32_t> argb;  // ARGB, scanline order
    int width = 0, height = 0;
    RasterImage() = default;
    RasterImage(int w, int h) : argb((size_t)w * h, 0), width(w), height(h) {}   // RasterImage.java:22-26
};

class FractalCompression {
public:
    static inline int blockgroesse = 8;          // FractalCompression.java:14
    static inline int widthKernel = 2;           // FractalCompression.java:15
    static inline int isometries = 1;            // extension (1 == the reference algorithm)
    static inline int device = 0;
    static inline std::vector<float> imageInfo;  // [N_r][3] = {i_local, a, b}  (FractalCompression.java:17,124)

    static bool isGreyScale(const RasterImage& input)   // FractalCompression.java:32-45
    {
        return check(fic_is_greyscale_argb(input.argb.data(), input.width, input.height)) == 1;
    }

    static inline float avgError = 0.0f;         // FractalCompression.java:20 -- never reset between decode calls
    static float getAvgError() { return avgError; }   // FractalCompression.java:22-24

    // FractalCompression.java:84-100: window origin {dy, dx} of candidate-centre `index` (host integer logic).
    static std::array<int, 2> generateKernel(int domainbloeckePerWidth, int domainbloeckePerHeight, int index)
    {
        int dy = index / domainbloeckePerWidth - widthKernel / 2;
        int dx = index % domainbloeckePerWidth - widthKernel / 2;
        if (dx < 0) dx = 0;
        if (dy < 0) dy = 0;
        if (dx + widthKernel >= domainbloeckePerWidth) dx = domainbloeckePerWidth - widthKernel;
        if (dy + widthKernel >= domainbloeckePerHeight) dy = domainbloeckePerHeight - widthKernel;
        return {dy, dx};
    }

    // FractalCompression.java:1142-1148
    static RasterImage generateGrayImage(int width, int height)
    {
        RasterImage image(width, height);
        for (auto& p : image.argb) p = (int32_t)0xff808080u;
        return image;
    }

    // FractalCompression.java:54-59.  Returns the collage image like the reference.
    static RasterImage encode(const RasterImage& input, std::ostream& out)
    {
        if (isGreyScale(input)) return encodeGrayScale(input, out);
        return encodeRGB(input, out);
    }

    // FractalCompression.java:171-219: joint-RGB search on the GPU, writeData(out, 1, ...), RGB collage back.
    static RasterImage encodeRGB(const RasterImage& input, std::ostream& out)
    {
        int Rw = 0, Rh = 0;
        check(fic_geometry(input.width, input.height, blockgroesse, &Rw, &Rh, nullptr, nullptr));
        const int nr = Rw * Rh;
        std::vector<int32_t> idx(nr), q((size_t)nr * 5);
        std::vector<float> a(nr), bR(nr), bG(nr), bB(nr);
        RasterImage collage(input.width, input.height);
        check(fic_encode_rgb_argb(input.argb.data(), input.width, input.height, blockgroesse, widthKernel, device, idx.data(),
                                  a.data(), bR.data(), bG.data(), bB.data(), q.data(), collage.argb.data()));
        std::vector<uint8_t> buf(20 + 4 * q.size());
        int64_t n = fic_write_run_rgb(q.data(), nr, input.width, input.height, blockgroesse, widthKernel, buf.data(),
                                      (int64_t)buf.size());
        if (n < 0) throw std::runtime_error(fic_last_error());
        out.write(reinterpret_cast<const char*>(buf.data()), n);
        out.flush();
        return collage;
    }

    // FractalCompression.java:547-553: reads the whole .run stream and dispatches on its first int.
    static RasterImage decode(std::istream& in)
    {
        std::vector<uint8_t> run((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (run.size() < 20) throw std::runtime_error("decode: stream shorter than its header (EOFException)");
        const bool rgb = (run[0] | run[1] | run[2] | run[3]) != 0;
        int w = 0, h = 0;
        auto be = [&](size_t o) { return (int)((run[o] << 24) | (run[o + 1] << 16) | (run[o + 2] << 8) | run[o + 3]); };
        RasterImage img(be(4) > 0 ? be(4) : 0, be(8) > 0 ? be(8) : 0);
        if (rgb) {
            check(fic_decode_rgb_run(run.data(), (int64_t)run.size(), device, img.argb.data(), (int64_t)img.argb.size(), &w, &h,
                                     &avgError, nullptr));
        } else {
            std::vector<uint8_t> g(img.argb.size());
            check(fic_decode_gray_run(run.data(), (int64_t)run.size(), device, g.data(), (int64_t)g.size(), &w, &h, &avgError,
                                      nullptr));
            for (size_t i = 0; i < g.size(); i++)
                img.argb[i] = (int32_t)(0xff000000u | ((uint32_t)g[i] << 16) | ((uint32_t)g[i] << 8) | g[i]);
        }
        return img;
    }

    // FractalCompression.java:109-162: search on the GPU, writeData to `out`, collage back.
    static RasterImage encodeGrayScale(const RasterImage& input, std::ostream& out)
    {
        fic_ctx* ctx = fic_ctx_create(device, input.width, input.height, blockgroesse, widthKernel, isometries, 1);
        if (!ctx) throw std::runtime_error(fic_last_error());
        Guard guard{ctx};
        check(fic_ctx_set_argb_host(ctx, input.argb.data()));
        check(fic_ctx_encode(ctx, 0, -1, nullptr));
        int Rw = 0, Rh = 0;
        check(fic_geometry(input.width, input.height, blockgroesse, &Rw, &Rh, nullptr, nullptr));
        const int nr = Rw * Rh;
        std::vector<int32_t> idx(nr), q((size_t)nr * 3);
        std::vector<float> a(nr), b(nr);
        check(fic_ctx_get_results_host(ctx, idx.data(), a.data(), b.data(), nullptr, q.data(), nullptr, nullptr));
        imageInfo.resize((size_t)nr * 3);
        for (int j = 0; j < nr; j++) {
            imageInfo[3 * j + 0] = (float)idx[j];
            imageInfo[3 * j + 1] = a[j];
            imageInfo[3 * j + 2] = b[j];
        }
        writeData(out, q, input.width, input.height);                 // FractalCompression.java:160
        RasterImage collage(input.width, input.height);               // FractalCompression.java:161
        check(fic_ctx_collage_host(ctx, collage.argb.data()));
        return collage;
    }

    // FractalCompression.java:230-246 (grey branch): big-endian header + rows.
    static void writeData(std::ostream& out, const std::vector<int32_t>& qrows, int width, int height)
    {
        std::vector<uint8_t> buf(20 + 4 * qrows.size());
        int64_t n = fic_write_run_gray(qrows.data(), (int)(qrows.size() / 3), width, height, blockgroesse, widthKernel,
                                       buf.data(), (int64_t)buf.size());
        if (n < 0) throw std::runtime_error(fic_last_error());
        out.write(reinterpret_cast<const char*>(buf.data()), n);
        out.flush();
    }

private:
    struct Guard {
        fic_ctx* c;
        ~Guard() { fic_ctx_destroy(c); }
    };
    static int check(int rc)
    {
        if (rc < 0) throw std::runtime_error(std::string("fic error ") + std::to_string(rc) + ": " + fic_last_error());
        return rc;
    }
};

}  // namespace bvk_ss19
