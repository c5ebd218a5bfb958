// Host check of the launch recorder's argument packing (tengine_amd/csrc/launch_rec.h: rec_pack): the explicit kernel-argument
// segment must lay every by-value argument out at its natural alignment, in order -- the layout the code object's metadata
// gives them (.args: .offset / .size) and direct.cc copies in front of the hidden arguments.  No device is touched.
// build: hipcc -std=c++17 -I tengine_amd/csrc rec_pack_check.cc -o rec_pack_check
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "launch_rec.h"

namespace tamd { thread_local std::vector<LaunchRec>* g_launch_rec = nullptr; }

struct Blob { const void* p; int a; float b; char c; };           // 8-aligned, 24 bytes with tail padding
struct Layout { char c; double d; int i; const void* p; short s; Blob b; float f; };   // what a kernel(char, double, int, ptr, short, Blob, float) sees

int main()
{
    std::vector<unsigned char> buf;
    const char c = 7; const double d = 2.5; const int i = -3; const void* p = &buf; const short s = 11; const Blob b{&buf, 4, 1.5f, 'x'}; const float f = 9.f;
    tamd::rec_pack(buf, c); tamd::rec_pack(buf, d); tamd::rec_pack(buf, i); tamd::rec_pack(buf, p); tamd::rec_pack(buf, s); tamd::rec_pack(buf, b); tamd::rec_pack(buf, f);
    int bad = 0;
    auto at = [&](size_t off, const void* v, size_t n, const char* what) {
        if (off + n > buf.size() || memcmp(buf.data() + off, v, n) != 0) { printf("MISMATCH %s at %zu\n", what, off); bad++; }
    };
    at(offsetof(Layout, c), &c, 1, "char"); at(offsetof(Layout, d), &d, 8, "double"); at(offsetof(Layout, i), &i, 4, "int");
    at(offsetof(Layout, p), &p, 8, "pointer"); at(offsetof(Layout, s), &s, 2, "short"); at(offsetof(Layout, b), &b, sizeof(Blob), "struct");
    at(offsetof(Layout, f), &f, 4, "float");
    if (buf.size() != offsetof(Layout, f) + 4) { printf("MISMATCH size %zu\n", buf.size()); bad++; }
    printf("packed %zu bytes, mismatches %d\n", buf.size(), bad);
    return bad != 0;
}
