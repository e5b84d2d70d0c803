package org.simdjson;

import java.lang.foreign.Arena;
import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.Linker;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.SymbolLookup;
import java.lang.invoke.MethodHandle;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/**
 * Downcall handles of libsjmi.so (include/sjmi.h), the MI355X engine behind SimdJsonParser.stage1 and
 * StringParser.parseString.  Foreign Function &amp; Memory API (final since JDK 22; the library needs JDK 24).
 * Heap arrays cross the boundary without a copy: every handle is linked with Linker.Option.critical(true), so
 * MemorySegment.ofArray(byte[] / int[] / long[]) is a legal argument.  The library is looked up through the system
 * property org.simdjson.sjmi (default: libsjmi.so on the loader path).
 *
 * One handle per C entry point that INTEGRATION.md binds; the C prototype is quoted above each.
 */
final class Sjmi {

    private static final Linker LINKER = Linker.nativeLinker();
    private static final SymbolLookup LIB =
            SymbolLookup.libraryLookup(System.getProperty("org.simdjson.sjmi", "libsjmi.so"), Arena.global());

    private static MethodHandle h(String name, FunctionDescriptor descriptor) {
        MemorySegment symbol = LIB.find(name).orElseThrow(() -> new UnsatisfiedLinkError("libsjmi.so: missing " + name));
        return LINKER.downcallHandle(symbol, descriptor, Linker.Option.critical(true));
    }

    // status word of stage 1 (SJMI_ST_*): the lowest set bit is thrown first, which is the reference's order of checks
    static final int ST_UTF8 = 1;
    static final int ST_UNCLOSED = 2;
    static final int ST_UNESCAPED = 4;
    static final int ST_CAPACITY = 0x100;
    static final int ST_INTERNAL = 0x200;
    static final int ST_HALO = 0x400;
    // sjmi_parse_document: the document needs the host walker (only with maxDepth > 1024)
    static final int WALK_NEEDS_HOST = -1;

    // int sjmi_create(sjmi_ctx** out, int device, uint64_t capacity_bytes)
    static final MethodHandle CREATE = h("sjmi_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_LONG));
    // void sjmi_destroy(sjmi_ctx* ctx)
    static final MethodHandle DESTROY = h("sjmi_destroy", FunctionDescriptor.ofVoid(ADDRESS));
    // const char* sjmi_last_error(const sjmi_ctx* ctx)
    static final MethodHandle LAST_ERROR = h("sjmi_last_error", FunctionDescriptor.of(ADDRESS, ADDRESS));

    // int sjmi_stage1(ctx, const uint8_t* buf, uint64_t len, uint32_t* indexes, uint64_t index_capacity,
    //                 uint64_t* count, uint32_t* status)
    static final MethodHandle STAGE1 = h("sjmi_stage1",
            FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS));
    // int sjmi_unescape(ctx, uint8_t* string_buffer, uint64_t string_capacity, uint64_t* total_bytes,
    //                   uint64_t* first_error_index, uint32_t* first_error_code)
    static final MethodHandle UNESCAPE = h("sjmi_unescape",
            FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS));
    // int sjmi_stage1_unescape(ctx, const uint8_t* buf, uint64_t len, uint32_t* indexes, uint64_t index_capacity,
    //                          uint64_t* count, uint32_t* status, uint8_t* string_buffer, uint64_t string_capacity,
    //                          uint64_t* total_bytes, uint64_t* first_error_index, uint32_t* first_error_code)
    static final MethodHandle STAGE1_UNESCAPE = h("sjmi_stage1_unescape",
            FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS,
                    ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS));

    // int sjmi_parse_document(ctx, const uint8_t* buf, uint64_t len, int max_depth, uint64_t* tape, uint64_t tape_capacity,
    //                         uint64_t* tape_len, uint8_t* string_buffer, uint64_t string_capacity, uint64_t* strings_len,
    //                         int32_t* error, uint32_t* stage1_status)
    static final MethodHandle PARSE_DOCUMENT = h("sjmi_parse_document",
            FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS,
                    ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS));

    // int sjmi_stage1_batch_isolated(ctx, const uint8_t* buf, uint64_t total_len, const uint64_t* doc_offsets, uint64_t n_docs,
    //                                uint32_t* indexes, uint64_t index_capacity, uint64_t* index_offsets,
    //                                uint32_t* doc_status, uint64_t* count, uint32_t* status)
    static final MethodHandle STAGE1_BATCH_ISOLATED = h("sjmi_stage1_batch_isolated",
            FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, JAVA_LONG, ADDRESS, JAVA_LONG, ADDRESS,
                    ADDRESS, ADDRESS, ADDRESS));
    // int sjmi_unescape_batch(ctx, uint8_t* string_buffer, uint64_t string_capacity, uint64_t* doc_string_offsets,
    //                         uint64_t* total_bytes, uint64_t* first_error_index, uint32_t* first_error_code)
    static final MethodHandle UNESCAPE_BATCH = h("sjmi_unescape_batch",
            FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS));

    // int sjmi_match_brackets(ctx, uint32_t* up, uint32_t* match, uint64_t capacity)
    static final MethodHandle MATCH_BRACKETS = h("sjmi_match_brackets",
            FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG));

    // int sjmi_stream_open(ctx, uint64_t max_chunk_bytes, uint64_t halo_bytes, sjmi_stream** out)
    static final MethodHandle STREAM_OPEN = h("sjmi_stream_open",
            FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS));
    // int sjmi_stream_push(sjmi_stream* s, const uint8_t* chunk, uint64_t len, int is_last, uint32_t* indexes,
    //                      uint64_t index_capacity, uint64_t* count, uint64_t* base, uint32_t* status)
    static final MethodHandle STREAM_PUSH = h("sjmi_stream_push",
            FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS));
    // void sjmi_stream_close(sjmi_stream* s)
    static final MethodHandle STREAM_CLOSE = h("sjmi_stream_close", FunctionDescriptor.ofVoid(ADDRESS));

    // int sjmi_host_register(ctx, void* ptr, uint64_t bytes) / int sjmi_host_unregister(ctx, void* ptr): off-heap parser buffers
    static final MethodHandle HOST_REGISTER = h("sjmi_host_register", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG));
    static final MethodHandle HOST_UNREGISTER = h("sjmi_host_unregister", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS));

    /** new SimdJsonParser(capacity, maxDepth): one engine context (device buffers + a HIP stream) per parser = per thread. */
    static MemorySegment create(int device, long capacityBytes) {
        try (Arena arena = Arena.ofConfined()) {
            MemorySegment out = arena.allocate(ADDRESS);
            int rc = (int) CREATE.invokeExact(out, device, capacityBytes);
            if (rc != 0) {
                throw new IllegalStateException("sjmi_create failed: " + rc);
            }
            return out.get(ADDRESS, 0);
        } catch (RuntimeException | Error e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** infrastructure failures (HIP, capacity) are never JSON verdicts: they surface as IllegalStateException */
    static void check(int rc, String call, MemorySegment ctx) {
        if (rc == 0) {
            return;
        }
        String detail = "";
        try {
            MemorySegment msg = (MemorySegment) LAST_ERROR.invokeExact(ctx);
            if (!msg.equals(MemorySegment.NULL)) {
                detail = ": " + msg.reinterpret(4096).getString(0);
            }
        } catch (Throwable ignored) {
            // keep the return code
        }
        throw new IllegalStateException(call + " failed: " + rc + detail);
    }

    /** the JsonParsingException of a stage-1 status word, in the reference's order (Utf8Validator.java:165-167,
     *  StructuralIndexer.java:297-302) */
    static void throwStage1(int status) {
        if ((status & ST_UTF8) != 0) {
            throw new JsonParsingException("The input is not valid UTF-8");
        }
        if ((status & ST_UNCLOSED) != 0) {
            throw new JsonParsingException("Unclosed string. A string is opened, but never closed.");
        }
        if ((status & ST_UNESCAPED) != 0) {
            throw new JsonParsingException("Unescaped characters. Within strings, there are characters that should be escaped.");
        }
    }

    private Sjmi() {
    }
}
