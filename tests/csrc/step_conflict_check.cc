// Host check of the dependency test behind the barrier-free launches of the direct path (tengine_amd/csrc/graph.h: Access,
// access_of, access_overlap, step_conflict): which steps may run beside each other.  No device is touched.
// build: hipcc -std=c++17 -I tengine_amd/csrc -I include step_conflict_check.cc -o step_conflict_check
#include <cstdio>

#include "graph.h"

using namespace tamd;

static int bad = 0;
#define EXPECT(c) do { if (!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); bad++; } } while (0)

static HTensor dense_u8(char* p, int n, int c, int h, int w)
{
    HTensor t; t.dtype = TAMD_DT_UINT8; t.dims = {n, c, h, w}; t.dptr = p; t.n = n; t.c = c; t.h = h; t.w = w; t.nchw_raw = true; t.cs = 0;
    return t;
}

int main()
{
    static char buf[1 << 20];
    // two dense NCHW tensors side by side, and one that overlaps the first
    HTensor a = dense_u8(buf, 2, 8, 4, 4), b = dense_u8(buf + 256, 2, 8, 4, 4), c = dense_u8(buf + 128, 2, 8, 4, 4);
    EXPECT(access_of(a).size == 256 && access_of(a).period == 0);
    EXPECT(!access_overlap(access_of(a), access_of(b)));
    EXPECT(access_overlap(access_of(a), access_of(c)) && access_overlap(access_of(c), access_of(b)));
    // uint8 concat-by-offset views: channel slices [0,3) and [3,8) of one 8-channel NCHW buffer are disjoint, [2,5) meets both
    HTensor v0 = dense_u8(buf, 2, 3, 4, 4), v1 = dense_u8(buf, 2, 5, 4, 4), v2 = dense_u8(buf, 2, 3, 4, 4);
    v0.is_view = v1.is_view = v2.is_view = true; v0.cs = v1.cs = v2.cs = 8; v0.c_off = 0; v1.c_off = 3; v2.c_off = 2;
    EXPECT(access_of(v0).period == 8 * 16 && access_of(v1).off == 3 * 16 && access_of(v1).len == 5 * 16 && access_of(v0).size == 256);
    EXPECT(!access_overlap(access_of(v0), access_of(v1)));
    EXPECT(access_overlap(access_of(v0), access_of(v2)) && access_overlap(access_of(v1), access_of(v2)));
    EXPECT(access_overlap(access_of(v0), access_of(a)));            // a slice against the whole buffer: conflict
    // int8 NHWC: cs bytes per pixel, a view owns channels [c_off, c_off + c) of every pixel
    HTensor n0; n0.dtype = TAMD_DT_INT8; n0.dims = {2, 16, 4, 4}; n0.dptr = buf; n0.n = 2; n0.c = 16; n0.h = 4; n0.w = 4; n0.cs = 32; n0.is_view = true; n0.c_off = 0;
    HTensor n1 = n0; n1.c_off = 16;
    HTensor whole = n0; whole.is_view = false; whole.c = 32; whole.dims = {2, 32, 4, 4};
    EXPECT(access_of(n0).size == 2u * 16 * 32 && access_of(n0).period == 32 && access_of(n1).off == 16);
    EXPECT(!access_overlap(access_of(n0), access_of(n1)) && access_overlap(access_of(n0), access_of(whole)));
    // a concat input's slot: bytes [off, off + len) of every out_img bytes
    Access s0, s1, s2;
    s0.base = s1.base = s2.base = buf; s0.size = s1.size = s2.size = 2 * 100; s0.period = s1.period = s2.period = 100;
    s0.off = 0; s0.len = 40; s1.off = 40; s1.len = 60; s2.off = 30; s2.len = 20;
    EXPECT(!access_overlap(s0, s1) && access_overlap(s0, s2) && access_overlap(s1, s2));
    // steps: RAW, WAR, WAW and the independent case (two head convolutions reading one map, writing their own tensors)
    Step conv_loc, conv_conf, cat_loc, next;
    HTensor src = dense_u8(buf + 4096, 2, 8, 4, 4), loc = dense_u8(buf + 8192, 2, 4, 4, 4), conf = dense_u8(buf + 12288, 2, 6, 4, 4), out = dense_u8(buf + 16384, 2, 20, 4, 4);
    conv_loc.deps = conv_conf.deps = cat_loc.deps = next.deps = true;
    conv_loc.rd = {access_of(src)}; conv_loc.wr = {access_of(loc)};
    conv_conf.rd = {access_of(src)}; conv_conf.wr = {access_of(conf)};
    cat_loc.rd = {access_of(loc)}; cat_loc.wr = {access_of(out)};
    next.rd = {access_of(conf)}; next.wr = {access_of(src)};         // overwrites what the convolutions read
    EXPECT(!step_conflict(conv_loc, conv_conf) && !step_conflict(conv_conf, conv_loc));       // read-read only
    EXPECT(step_conflict(cat_loc, conv_loc) && step_conflict(conv_loc, cat_loc));             // RAW (either argument order)
    EXPECT(!step_conflict(cat_loc, conv_conf));
    EXPECT(step_conflict(next, conv_loc));                                                    // WAR
    EXPECT(step_conflict(next, conv_conf));                                                   // RAW + WAR
    Step again = conv_loc;
    EXPECT(step_conflict(again, conv_loc));                                                   // WAW
    printf("step_conflict_check: %d failures\n", bad);
    return bad != 0;
}
