// jpeg_to_raw in.jpg out.raw — decodes with tools/jpeg_decode.hpp and writes "w h channels\n" + the interleaved bytes
// (test helper for tests/test_jpeg_decode.py).
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>
#include "jpeg_decode.hpp"
int main(int argc, char** argv) {
    if (argc != 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    svo_jpeg::Image im = svo_jpeg::decode(d.data(), d.size());
    if (!im.ok) { std::fprintf(stderr, "decode failed\n"); return 1; }
    FILE* o = std::fopen(argv[2], "wb");
    std::fprintf(o, "%d %d %d\n", im.w, im.h, im.channels);
    std::fwrite(im.px.data(), 1, im.px.size(), o);
    std::fclose(o);
    return 0;
}
