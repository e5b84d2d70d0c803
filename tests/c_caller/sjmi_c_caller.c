/* sjmi_c_caller.c -- a C host of libsjmi.so that does what the Java binding of INTEGRATION.md does, with nothing but
 * dlopen / dlsym (= SymbolLookup.libraryLookup + Linker.downcallHandle) and plain pointers and sizes:
 *
 *   replay <file>   the FFM call sequence of INTEGRATION.md sections 2-3 and 6: sjmi_create; sjmi_stage1_unescape on the
 *                   caller's unpadded byte[] (SimdJsonParser.stage1); setWriteIdx(count) + the sentinel; a visitString loop
 *                   over every '"' structural that reads the record lengths the GPU wrote (TapeBuilder.visitString); then
 *                   sjmi_parse_document (tape from the device) and sjmi_destroy.  Prints counts and FNV-1a hashes that
 *                   tests/test_gpu_c_caller.py compares with the oracle's.
 *   engine <file>   what java/org/simdjson/Sjmi.java's Engine does since round 5 (ordinary, non-critical downcalls need off-heap
 *                   segments): three 64-byte-aligned native buffers (input + padding, index array, string buffer), each
 *                   sjmi_host_register-ed, the input one also handed to sjmi_set_input_staging; per parse: memcpy of the
 *                   document into the input buffer, sjmi_stage1_unescape on THOSE buffers (the kernels write indexes and
 *                   records zero-copy), copies out; twice (the second call uses the cached device views); then
 *                   sjmi_host_unregister x3 + sjmi_destroy (Engine.close / the Cleaner action).  Same first output line as replay.
 *   guard <file>    the visibility contract of include/sjmi.h ("bytes >= len are never read from buf": what lets Java hand
 *                   over an unpadded byte[]): the document is placed so that byte `len` is the FIRST BYTE OF A PROT_NONE PAGE
 *                   and sjmi_stage1, sjmi_stage1_unescape, sjmi_parser_parse (both placements of stage 2) and sjmi_stream_push
 *                   are called on it for len in {0, 1, 63, 64, 65, 4095, 4096, size of <file>}.  A read past len is a SIGSEGV.
 *
 * Test infrastructure (built by __graft_entry__.build() with gcc; no HIP headers, no torch). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include "sjmi.h"

#define SYM(name) static __typeof__(&name) p_##name
SYM(sjmi_create); SYM(sjmi_destroy); SYM(sjmi_last_error); SYM(sjmi_stage1); SYM(sjmi_stage1_unescape); SYM(sjmi_parse_document);
SYM(sjmi_parser_create); SYM(sjmi_parser_destroy); SYM(sjmi_parser_parse); SYM(sjmi_parser_set_gpu_walk); SYM(sjmi_parser_last_message);
SYM(sjmi_stream_open); SYM(sjmi_stream_push); SYM(sjmi_stream_close);
SYM(sjmi_host_register); SYM(sjmi_host_unregister); SYM(sjmi_set_input_staging);

static void* lib;
#define LOAD(name) do { p_##name = (__typeof__(p_##name))dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); exit(2); } } while (0)

static uint64_t fnv(const void* data, size_t n, uint64_t h) {
    const uint8_t* p = (const uint8_t*)data;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}
#define FNV0 0xCBF29CE484222325ull

static uint8_t* read_file(const char* path, size_t* n) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    *n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* b = (uint8_t*)malloc(*n ? *n : 1);
    if (fread(b, 1, *n, f) != *n) { perror("fread"); exit(2); }
    fclose(f);
    return b;
}

static int replay(const char* path) {
    size_t n;
    uint8_t* doc = read_file(path, &n);   /* exactly n bytes: the "byte[] buffer" of SimdJsonParser.parse, no padding */
    sjmi_ctx* ctx = NULL;
    if (p_sjmi_create(&ctx, 0, n + 64) != SJMI_OK) { fprintf(stderr, "sjmi_create failed\n"); return 1; }
    const uint64_t cap = n + 2;
    uint32_t* indexes = (uint32_t*)malloc(cap * 4);             /* BitIndexes.indexes */
    const uint64_t sb_cap = n + 4 * (n / 2 + 2) + 64;
    uint8_t* sb = (uint8_t*)malloc(sb_cap);                     /* the parser's stringBuffer */
    uint64_t count = 0, total = 0, fei = 0;
    uint32_t status = 0, fec = 0;
    int rc = p_sjmi_stage1_unescape(ctx, doc, n, indexes, cap, &count, &status, sb, sb_cap, &total, &fei, &fec);
    if (rc != SJMI_OK) { fprintf(stderr, "sjmi_stage1_unescape: %d %s\n", rc, p_sjmi_last_error(ctx)); return 1; }
    if (indexes[count] != 0) { fprintf(stderr, "sentinel missing\n"); return 1; }  /* BitIndexes.finish */
    /* TapeBuilder.visitString for every string in structural order: the record sits at the running offset */
    uint64_t sbi = 0, nstr = 0, bad = 0;
    for (uint64_t i = 0; i < count && status == 0; ++i) {
        if (doc[indexes[i]] != '"') continue;
        const uint32_t len = ((uint32_t)sb[sbi] << 24) | ((uint32_t)sb[sbi + 1] << 16) | ((uint32_t)sb[sbi + 2] << 8) | sb[sbi + 3];
        if ((len & 0xFFFFFF00u) == 0xFFFFFF00u) { ++bad; break; }
        sbi += 4 + len;
        ++nstr;
    }
    printf("stage1 count=%llu status=%u idxhash=%016llx strings=%llu string_bytes=%llu walked_bytes=%llu sbhash=%016llx bad=%llu\n",
           (unsigned long long)count, status, (unsigned long long)fnv(indexes, count * 4, FNV0), (unsigned long long)nstr,
           (unsigned long long)total, (unsigned long long)sbi, (unsigned long long)fnv(sb, total, FNV0), (unsigned long long)bad);
    /* the tape from the device */
    const uint64_t tcap = 2 * count + 16;
    uint64_t* tape = (uint64_t*)malloc(tcap * 8);
    uint64_t tlen = 0, slen = 0;
    int32_t err = 0;
    uint32_t st1 = 0;
    rc = p_sjmi_parse_document(ctx, doc, n, 1024, tape, tcap, &tlen, sb, sb_cap, &slen, &err, &st1);
    if (rc != SJMI_OK) { fprintf(stderr, "sjmi_parse_document: %d %s\n", rc, p_sjmi_last_error(ctx)); return 1; }
    printf("document error=%d stage1_status=%u tape_len=%llu tapehash=%016llx strings_len=%llu\n", err, st1, (unsigned long long)tlen,
           (unsigned long long)fnv(tape, tlen * 8, FNV0), (unsigned long long)slen);
    p_sjmi_destroy(ctx);
    free(tape); free(sb); free(indexes); free(doc);
    return 0;
}

static int engine(const char* path) {
    size_t n;
    uint8_t* doc = read_file(path, &n);
    const uint64_t capacity = ((n + 2) * 4 + 4095) / 4096 * 4096;   /* new SimdJsonParser(capacity, maxDepth) */
    sjmi_ctx* ctx = NULL;
    if (p_sjmi_create(&ctx, 0, capacity) != SJMI_OK) { fprintf(stderr, "sjmi_create failed\n"); return 1; }
    void *in = NULL, *indexes = NULL, *strings = NULL;             /* Arena.ofShared().allocate(bytes, 64) */
    const uint64_t in_bytes = capacity + 64, idx_bytes = 4 * capacity, sb_bytes = capacity;
    if (posix_memalign(&in, 4096, in_bytes) || posix_memalign(&indexes, 4096, idx_bytes) || posix_memalign(&strings, 4096, sb_bytes)) return 1;
    void* const pinned[3] = {in, indexes, strings};
    const uint64_t pinned_bytes[3] = {in_bytes, idx_bytes, sb_bytes};
    for (int i = 0; i < 3; ++i)
        if (p_sjmi_host_register(ctx, pinned[i], pinned_bytes[i]) != SJMI_OK) { fprintf(stderr, "sjmi_host_register: %s\n", p_sjmi_last_error(ctx)); return 1; }
    if (p_sjmi_set_input_staging(ctx, in, in_bytes) != SJMI_OK) { fprintf(stderr, "sjmi_set_input_staging failed\n"); return 1; }
    uint32_t* heap_idx = (uint32_t*)malloc((n + 2) * 4);              /* BitIndexes.indexes (heap int[]) */
    uint8_t* heap_sb = (uint8_t*)malloc(capacity);                    /* the parser's stringBuffer (heap byte[]) */
    for (int pass = 0; pass < 2; ++pass) {
        uint64_t out[5] = {0, 0, 0, 0, 0};                            /* count | status | total | firstErrorIndex | firstErrorCode */
        memset(indexes, 0xEE, idx_bytes);
        memcpy(in, doc, n);                                           /* MemorySegment.copy(buffer, 0, in, JAVA_BYTE, 0, length) */
        int rc = p_sjmi_stage1_unescape(ctx, (const uint8_t*)in, n, (uint32_t*)indexes, capacity, &out[0], (uint32_t*)&out[1],
                                        (uint8_t*)strings, capacity, &out[2], &out[3], (uint32_t*)&out[4]);
        if (rc != SJMI_OK) { fprintf(stderr, "sjmi_stage1_unescape: %d %s\n", rc, p_sjmi_last_error(ctx)); return 1; }
        const uint64_t count = out[0], total = out[2];
        const uint32_t status = (uint32_t)out[1];
        memcpy(heap_idx, indexes, (count + 1) * 4);                   /* MemorySegment.copy(indexes, JAVA_INT, 0, array, 0, count + 1) */
        if (heap_idx[count] != 0) { fprintf(stderr, "sentinel missing\n"); return 1; }
        memcpy(heap_sb, strings, total);
        uint64_t sbi = 0, nstr = 0, bad = 0;
        for (uint64_t i = 0; i < count && status == 0; ++i) {
            if (doc[heap_idx[i]] != '"') continue;
            const uint32_t len = ((uint32_t)heap_sb[sbi] << 24) | ((uint32_t)heap_sb[sbi + 1] << 16) | ((uint32_t)heap_sb[sbi + 2] << 8) | heap_sb[sbi + 3];
            if ((len & 0xFFFFFF00u) == 0xFFFFFF00u) { ++bad; break; }
            sbi += 4 + len;
            ++nstr;
        }
        printf("stage1 count=%llu status=%u idxhash=%016llx strings=%llu string_bytes=%llu walked_bytes=%llu sbhash=%016llx bad=%llu\n",
               (unsigned long long)count, status, (unsigned long long)fnv(heap_idx, count * 4, FNV0), (unsigned long long)nstr,
               (unsigned long long)total, (unsigned long long)sbi, (unsigned long long)fnv(heap_sb, total, FNV0), (unsigned long long)bad);
    }
    for (int i = 0; i < 3; ++i)
        if (p_sjmi_host_unregister(ctx, pinned[i]) != SJMI_OK) { fprintf(stderr, "sjmi_host_unregister failed\n"); return 1; }
    p_sjmi_destroy(ctx);
    free(heap_sb); free(heap_idx); free(strings); free(indexes); free(in); free(doc);
    return 0;
}

/* `len` bytes whose successor byte is the first byte of a PROT_NONE page */
static uint8_t* guarded(size_t len, void** map, size_t* map_len) {
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    const size_t body = (len + page - 1) / page * page;
    *map_len = body + 2 * page;
    uint8_t* m = (uint8_t*)mmap(NULL, *map_len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) { perror("mmap"); exit(2); }
    if (mprotect(m + page + body, page, PROT_NONE) != 0) { perror("mprotect"); exit(2); }
    if (mprotect(m, page, PROT_NONE) != 0) { perror("mprotect"); exit(2); }   /* (and the page in front, for good measure) */
    *map = m;
    return m + page + body - len;
}

static int guard(const char* path) {
    size_t fn;
    uint8_t* file = read_file(path, &fn);
    const size_t lens[] = {0, 1, 63, 64, 65, 4095, 4096, fn};
    sjmi_ctx* ctx = NULL;
    if (p_sjmi_create(&ctx, 0, fn + 4096) != SJMI_OK) { fprintf(stderr, "sjmi_create failed\n"); return 1; }
    sjmi_parser* parser = NULL;
    if (p_sjmi_parser_create(&parser, (int)(fn + 4096), 1024, 0) != SJMI_OK) { fprintf(stderr, "sjmi_parser_create failed\n"); return 1; }
    sjmi_stream* stream = NULL;
    for (size_t li = 0; li < sizeof lens / sizeof lens[0]; ++li) {
        const size_t len = lens[li];
        void* map;
        size_t map_len;
        uint8_t* buf = guarded(len, &map, &map_len);
        uint64_t want = 1;
        if (len == fn) memcpy(buf, file, fn);               /* the reference fixture itself */
        else if (len == 1) buf[0] = '7';
        else if (len >= 2) { memset(buf, 'a', len); buf[0] = '"'; buf[len - 1] = '"'; }   /* one string filling the buffer */
        if (len == 0) want = 0;
        const uint64_t cap = len + 2, sb_cap = len + 4 * (len / 2 + 2) + 64;
        uint32_t* indexes = (uint32_t*)malloc(cap * 4);
        uint8_t* sb = (uint8_t*)malloc(sb_cap);
        uint64_t count = 0, total = 0, fei = 0, c2 = 0;
        uint32_t status = 0, fec = 0, s2 = 0;
        int rc = p_sjmi_stage1(ctx, buf, len, indexes, cap, &count, &status);
        if (rc != SJMI_OK || status != 0) { fprintf(stderr, "len %zu: sjmi_stage1 rc=%d status=%u\n", len, rc, status); return 1; }
        rc = p_sjmi_stage1_unescape(ctx, buf, len, indexes, cap, &c2, &s2, sb, sb_cap, &total, &fei, &fec);
        if (rc != SJMI_OK || s2 != 0 || c2 != count) { fprintf(stderr, "len %zu: sjmi_stage1_unescape rc=%d\n", len, rc); return 1; }
        if (len != fn && count != want) { fprintf(stderr, "len %zu: count %llu\n", len, (unsigned long long)count); return 1; }
        if (len != fn && len >= 2 && total != 4 + len - 2) { fprintf(stderr, "len %zu: string bytes %llu\n", len, (unsigned long long)total); return 1; }
        for (int gpu_walk = 0; gpu_walk <= 1; ++gpu_walk) {
            const uint64_t* tape = NULL;
            const uint8_t* strings = NULL;
            uint64_t tlen = 0, slen = 0, epos = 0;
            p_sjmi_parser_set_gpu_walk(parser, gpu_walk);
            rc = p_sjmi_parser_parse(parser, buf, len, &tape, &tlen, &strings, &slen, &epos);
            const int expect = len == 0 ? SJMI_E_NO_STRUCTURAL : 0;  /* JsonIterator.java:27-29 */
            if (rc != expect) { fprintf(stderr, "len %zu: sjmi_parser_parse(gpu_walk=%d) rc=%d (%s)\n", len, gpu_walk, rc, p_sjmi_parser_last_message(parser)); return 1; }
        }
        if (p_sjmi_stream_open(ctx, len ? len : 64, 0, &stream) != SJMI_OK) { fprintf(stderr, "sjmi_stream_open failed\n"); return 1; }
        uint64_t sc = 0, base = 0;
        uint32_t ss = 0;
        rc = p_sjmi_stream_push(stream, buf, len, 1, indexes, cap, &sc, &base, &ss);
        if (rc != SJMI_OK || ss != 0 || sc != count) { fprintf(stderr, "len %zu: sjmi_stream_push rc=%d status=%u count=%llu\n", len, rc, ss, (unsigned long long)sc); return 1; }
        p_sjmi_stream_close(stream);
        printf("len %zu ok count=%llu\n", len, (unsigned long long)count);
        free(sb); free(indexes);
        munmap(map, map_len);
    }
    p_sjmi_parser_destroy(parser);
    p_sjmi_destroy(ctx);
    free(file);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s <libsjmi.so> replay|engine|guard <json file>\n", argv[0]); return 2; }
    lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(sjmi_create); LOAD(sjmi_destroy); LOAD(sjmi_last_error); LOAD(sjmi_stage1); LOAD(sjmi_stage1_unescape); LOAD(sjmi_parse_document);
    LOAD(sjmi_parser_create); LOAD(sjmi_parser_destroy); LOAD(sjmi_parser_parse); LOAD(sjmi_parser_set_gpu_walk); LOAD(sjmi_parser_last_message);
    LOAD(sjmi_stream_open); LOAD(sjmi_stream_push); LOAD(sjmi_stream_close);
    LOAD(sjmi_host_register); LOAD(sjmi_host_unregister); LOAD(sjmi_set_input_staging);
    if (!strcmp(argv[2], "replay")) return replay(argv[3]);
    if (!strcmp(argv[2], "engine")) return engine(argv[3]);
    if (!strcmp(argv[2], "guard")) return guard(argv[3]);
    return 2;
}
