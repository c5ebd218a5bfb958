// INTEGRATION.md section F compiled for real: several GPUs from C, no PyTorch.  One process per GPU:
//
//     rccl_gather <rank> <world> <id_file> <tmfile> <total_images> [steps]
//
//   rank 0 reads the tmfile from disk and creates the RCCL unique id (written to <id_file>, the other ranks poll for it -- the
//   only out-of-band step, what an MPI launcher or torchrun's store does elsewhere);
//   ncclBroadcast of the byte count, then of the raw tmfile bytes (device buffers) -> every rank loads the SAME bytes with the
//   native loader (tamd_graph_load_tm2 == create_graph(ctx, "tengine:m", buf, size), c_api.c:399-421) and checks a checksum of
//   what arrived against rank 0's (second broadcast);
//   static contiguous image shards (total/world, remainder to the low ranks), seeded input per GLOBAL image index;
//   `steps` passes with no collective (independent images), then ONE ncclAllGather per graph output, shards padded to the largest;
//   rank 0 prints a line per output: bytes per image, FNV-1a of the gathered results in global image order.
// tests/test_gpu_rccl_c.py runs it with world 1 on the GPU box and compares the hashes with the Python binding's results.
// build: tengine_amd/build.py build_harness() -> tengine_amd/lib/rccl_gather.bin (hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include ... -ltengine_amd -lrccl -Wl,-rpath,'$ORIGIN')
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <vector>

#include "tengine_amd.h"

#define HIPOK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #e, hipGetErrorString(r_)); exit(2); } } while (0)
#define NCCLOK(e) do { ncclResult_t r_ = (e); if (r_ != ncclSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #e, ncclGetErrorString(r_)); exit(3); } } while (0)
#define TAMDOK(e) do { if ((e) != 0) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #e, tamd_last_error()); exit(4); } } while (0)
static int g_rank = 0;

static uint64_t fnv1a(const void* p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

static void shard_range(int total, int world, int rank, int* start, int* count)      // tengine_amd/dist.py: shard_range
{
    const int base = total / world, rem = total % world;
    *count = base + (rank < rem ? 1 : 0);
    *start = rank * base + (rank < rem ? rank : rem);
}

int main(int argc, char** argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s rank world id_file tmfile total_images [steps]\n", argv[0]); return 1; }
    const int rank = atoi(argv[1]), world = atoi(argv[2]), total = atoi(argv[5]), steps = argc > 6 ? atoi(argv[6]) : 3;
    const char *id_file = argv[3], *tm_path = argv[4];
    g_rank = rank;
    int ndev = 0;
    HIPOK(hipGetDeviceCount(&ndev));
    const int dev = rank % (ndev > 0 ? ndev : 1);             // world 1 / a one-GPU box: every rank on device 0
    HIPOK(hipSetDevice(dev));

    // ---- RCCL communicator: the unique id travels through a file ---------------------------------------------------------
    ncclUniqueId id;
    if (rank == 0) {
        NCCLOK(ncclGetUniqueId(&id));
        char tmp[1024];
        snprintf(tmp, sizeof(tmp), "%s.tmp", id_file);
        FILE* f = fopen(tmp, "wb");
        if (!f || fwrite(&id, sizeof(id), 1, f) != 1) { fprintf(stderr, "cannot write %s\n", tmp); return 1; }
        fclose(f);
        rename(tmp, id_file);
    } else {
        FILE* f = nullptr;
        for (int tries = 0; tries < 600 && !(f = fopen(id_file, "rb")); tries++) usleep(100000);
        if (!f || fread(&id, sizeof(id), 1, f) != 1) { fprintf(stderr, "rank %d: no unique id in %s\n", rank, id_file); return 1; }
        fclose(f);
    }
    ncclComm_t comm;
    NCCLOK(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t st;
    HIPOK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

    // ---- the model: rank 0 has the file, everybody gets the bytes -------------------------------------------------------------
    std::vector<unsigned char> tm;
    unsigned long long nbytes = 0;
    if (rank == 0) {
        FILE* f = fopen(tm_path, "rb");
        if (!f) { fprintf(stderr, "cannot read %s\n", tm_path); return 1; }
        fseek(f, 0, SEEK_END); nbytes = (unsigned long long)ftell(f); fseek(f, 0, SEEK_SET);
        tm.resize(nbytes);
        if (fread(tm.data(), 1, nbytes, f) != nbytes) { fprintf(stderr, "short read of %s\n", tm_path); return 1; }
        fclose(f);
    }
    unsigned long long* d_n;
    HIPOK(hipMalloc(&d_n, 16));
    HIPOK(hipMemcpy(d_n, &nbytes, 8, hipMemcpyHostToDevice));
    NCCLOK(ncclBroadcast(d_n, d_n, 1, ncclUint64, 0, comm, st));
    HIPOK(hipStreamSynchronize(st));
    HIPOK(hipMemcpy(&nbytes, d_n, 8, hipMemcpyDeviceToHost));
    unsigned char* d_tm;
    HIPOK(hipMalloc(&d_tm, nbytes));
    if (rank == 0) HIPOK(hipMemcpy(d_tm, tm.data(), nbytes, hipMemcpyHostToDevice));
    NCCLOK(ncclBroadcast(d_tm, d_tm, nbytes, ncclUint8, 0, comm, st));
    HIPOK(hipStreamSynchronize(st));
    tm.resize(nbytes);
    HIPOK(hipMemcpy(tm.data(), d_tm, nbytes, hipMemcpyDeviceToHost));
    unsigned long long h_mine = fnv1a(tm.data(), nbytes), h_root = h_mine;       // integrity: every rank must hold the same model
    HIPOK(hipMemcpy(d_n, &h_root, 8, hipMemcpyHostToDevice));
    NCCLOK(ncclBroadcast(d_n, d_n, 1, ncclUint64, 0, comm, st));
    HIPOK(hipStreamSynchronize(st));
    HIPOK(hipMemcpy(&h_root, d_n, 8, hipMemcpyDeviceToHost));
    if (h_root != h_mine) { fprintf(stderr, "rank %d: tmfile broadcast corrupted\n", rank); return 5; }

    // ---- this rank's shard ---------------------------------------------------------------------------------------------------
    int start, count, max_count, s0;
    shard_range(total, world, rank, &start, &count);
    shard_range(total, world, 0, &s0, &max_count);
    if (count == 0) { fprintf(stderr, "rank %d has no images\n", rank); return 1; }
    tamd_graph* g = tamd_graph_load_tm2(tm.data(), nbytes);
    if (!g) { fprintf(stderr, "rank %d: tamd_graph_load_tm2: %s\n", rank, tamd_last_error()); return 4; }
    TAMDOK(tamd_graph_set_batch(g, count));
    tamd_options opt;
    memset(&opt, 0, sizeof(opt));
    opt.dev_name = "HIP"; opt.size = (int)sizeof(opt); opt.gpu_index = dev; opt.use_hip_graph = 1; opt.direct_dispatch = 1;
    TAMDOK(tamd_graph_prerun(g, &opt));
    int dims[8], dtype = 0;
    const int nd = tamd_graph_input_desc(g, 0, dims, &dtype);
    size_t per_image = 1;
    for (int i = 1; i < nd; i++) per_image *= (size_t)dims[i];
    if (dtype == TAMD_DT_FP32) per_image *= 4;
    std::vector<unsigned char> x((size_t)count * per_image);
    for (int i = 0; i < count; i++) {                    // image (start + i) of the GLOBAL batch: the same bytes whatever the world size
        unsigned lcg = 0x5EED0000u + (unsigned)(start + i);
        for (size_t k = 0; k < per_image; k++) { lcg = lcg * 1664525u + 1013904223u; x[(size_t)i * per_image + k] = (unsigned char)(lcg >> 24); }
    }
    TAMDOK(tamd_graph_set_input(g, 0, x.data(), x.size()));
    TAMDOK(tamd_graph_upload_inputs(g));
    for (int k = 0; k < steps; k++) TAMDOK(tamd_graph_launch(g));        // no collective: the outputs stay in this rank's HBM
    TAMDOK(tamd_graph_sync(g));                                          // direct dispatch: the passes are not on the HIP stream

    // ---- results together: one all-gather per output ------------------------------------------------------------------------
    const int nout = tamd_graph_output_num(g);
    for (int o = 0; o < nout; o++) {
        void* dptr; size_t bytes;
        TAMDOK(tamd_graph_output_device(g, o, &dptr, &bytes));
        const size_t out_per_image = bytes / (size_t)count, slot = out_per_image * (size_t)max_count;
        unsigned char *d_slot, *d_all;
        HIPOK(hipMalloc(&d_slot, slot)); HIPOK(hipMalloc(&d_all, slot * world));
        HIPOK(hipMemsetAsync(d_slot, 0, slot, st));
        HIPOK(hipMemcpyAsync(d_slot, dptr, bytes, hipMemcpyDeviceToDevice, st));
        NCCLOK(ncclAllGather(d_slot, d_all, slot, ncclUint8, comm, st));
        HIPOK(hipStreamSynchronize(st));
        if (rank == 0) {
            std::vector<unsigned char> all(slot * world);
            HIPOK(hipMemcpy(all.data(), d_all, all.size(), hipMemcpyDeviceToHost));
            uint64_t h = 1469598103934665603ull;
            for (int r = 0; r < world; r++) {            // trim the padding: global image order
                int rs, rc;
                shard_range(total, world, r, &rs, &rc);
                h = fnv1a(all.data() + (size_t)r * slot, out_per_image * (size_t)rc, h);
            }
            printf("output %d bytes_per_image %zu images %d fnv1a %016llx\n", o, out_per_image, total, (unsigned long long)h);
        }
        HIPOK(hipFree(d_slot)); HIPOK(hipFree(d_all));
    }
    if (rank == 0) printf("ok world %d total %d steps %d direct_packets %d tmfile_bytes %llu\n", world, total, steps, tamd_graph_direct_packets(g), nbytes);
    tamd_graph_destroy(g);
    NCCLOK(ncclCommDestroy(comm));
    return 0;
}
