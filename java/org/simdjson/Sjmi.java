package org.simdjson;

import java.lang.foreign.Arena;
import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.Linker;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.SymbolLookup;
import java.lang.invoke.MethodHandle;
import java.lang.ref.Cleaner;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_BYTE;
import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/**
 * Downcall handles of libsjmi.so (include/sjmi.h), the MI355X engine behind SimdJsonParser.stage1 and
 * StringParser.parseString.  Foreign Function &amp; Memory API (final since JDK 22; the library needs JDK 24).
 * The library is looked up through the system property org.simdjson.sjmi (default: libsjmi.so on the loader path).
 *
 * <p><b>Which calls are critical.</b>  A downcall linked with Linker.Option.critical(true) never leaves the Java thread
 * state: while it runs, no safepoint can be reached, so a garbage collection anywhere in the JVM waits for it.  The engine's
 * entry points upload, launch kernels and synchronise with the GPU (sjmi_create allocates device memory; a large document
 * spawns a copy thread) -- they are linked as ORDINARY downcalls ({@link #h}).  Ordinary downcalls cannot take heap
 * segments, so the parser keeps its native-side buffers OFF-HEAP ({@link Engine}: input, index array, string buffer, the
 * small out-words), page-locks them once with sjmi_host_register -- which also switches the engine to its zero-copy
 * outputs: the kernels store indexes and string records straight into those segments -- and copies between them and the
 * reference's heap arrays on the Java side.  Only {@link #LAST_ERROR} (returns a pointer, touches nothing) is critical.
 *
 * <p>One handle per C entry point that INTEGRATION.md binds; the C prototype is quoted above each.
 */
final class Sjmi {

    private static final Linker LINKER = Linker.nativeLinker();
    private static final SymbolLookup LIB =
            SymbolLookup.libraryLookup(System.getProperty("org.simdjson.sjmi", "libsjmi.so"), Arena.global());

    private static MemorySegment symbol(String name) {
        return LIB.find(name).orElseThrow(() -> new UnsatisfiedLinkError("libsjmi.so: missing " + name));
    }

    /** an ordinary downcall: the thread is in native state for its duration, safepoints and GC proceed */
    private static MethodHandle h(String name, FunctionDescriptor descriptor) {
        return LINKER.downcallHandle(symbol(name), descriptor);
    }

    /** a critical downcall: only for calls that return at once and never block */
    private static MethodHandle critical(String name, FunctionDescriptor descriptor) {
        return LINKER.downcallHandle(symbol(name), descriptor, Linker.Option.critical(false));
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
    static final MethodHandle LAST_ERROR = critical("sjmi_last_error", FunctionDescriptor.of(ADDRESS, ADDRESS));

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
    // int sjmi_set_input_staging(ctx, void* pinned, uint64_t bytes): the parser's page-locked input segment IS the upload's source
    static final MethodHandle SET_INPUT_STAGING = h("sjmi_set_input_staging", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG));

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

    /**
     * What one SimdJsonParser owns on the native side: the engine context and four off-heap, page-locked segments.
     * Lifetime: {@link #close()} (SimdJsonParser implements AutoCloseable) or, for parsers that are simply dropped, a
     * {@link Cleaner} action -- either way sjmi_host_unregister for every segment, sjmi_destroy, then the arena (a context
     * holds device memory sized for the parser's capacity and a HIP stream: leaking one per collected parser is not an option).
     */
    static final class Engine implements AutoCloseable {

        private static final Cleaner CLEANER = Cleaner.create();

        /** the cleaning action: must not reference the Engine (or the parser) it cleans up after */
        private static final class Native implements Runnable {
            private final Arena arena;
            private final MemorySegment ctx;
            private final MemorySegment[] pinned;

            Native(Arena arena, MemorySegment ctx, MemorySegment[] pinned) {
                this.arena = arena;
                this.ctx = ctx;
                this.pinned = pinned;
            }

            @Override
            public void run() {
                try {
                    for (MemorySegment p : pinned) {
                        int ignored = (int) HOST_UNREGISTER.invokeExact(ctx, p);
                    }
                    DESTROY.invokeExact(ctx);
                } catch (Throwable t) {
                    // (nothing sensible to do from a cleaner thread)
                } finally {
                    arena.close();
                }
            }
        }

        private final MemorySegment ctx;
        private final MemorySegment in;        // capacity + padding bytes: padIfNeeded's copy lands here, the GPU reads it by DMA
        private final MemorySegment indexes;   // capacity ints: written by k_stage1 over PCIe (zero-copy), sentinel included
        private final MemorySegment strings;   // capacity bytes: written by k_strings
        private final MemorySegment out;       // count | status | total | firstErrorIndex | firstErrorCode, 8 bytes apart
        private final long capacity;
        private final Cleaner.Cleanable cleanable;

        Engine(Object owner, int device, int capacity, int padding) {
            this.capacity = capacity;
            Arena arena = Arena.ofShared();
            MemorySegment c = null;
            // the segments page-locked so far: a failure half way through must unregister exactly these before the arena frees
            // their memory (memory freed while still registered leaks the pin and leaves a stale registration behind)
            java.util.ArrayList<MemorySegment> registered = new java.util.ArrayList<>(3);
            try {
                c = create(device, capacity);
                in = arena.allocate((long) capacity + padding, 64);
                indexes = arena.allocate(4L * capacity, 64);
                strings = arena.allocate(capacity, 64);
                out = arena.allocate(64, 8);
                MemorySegment[] pinned = {in, indexes, strings};
                for (MemorySegment p : pinned) {
                    check((int) HOST_REGISTER.invokeExact(c, p, p.byteSize()), "sjmi_host_register", c);
                    registered.add(p);
                }
                check((int) SET_INPUT_STAGING.invokeExact(c, in, in.byteSize()), "sjmi_set_input_staging", c);
                ctx = c;
                cleanable = CLEANER.register(owner, new Native(arena, c, pinned));
            } catch (RuntimeException | Error e) {
                destroyQuietly(c, arena, registered);
                throw e;
            } catch (Throwable t) {
                destroyQuietly(c, arena, registered);
                throw new IllegalStateException(t);
            }
        }

        private static void destroyQuietly(MemorySegment c, Arena arena, java.util.List<MemorySegment> registered) {
            try {
                if (c != null) {
                    // (the partially built state goes through the same action the Cleaner would run for a complete one)
                    new Native(arena, c, registered.toArray(new MemorySegment[0])).run();
                    return;
                }
            } catch (Throwable ignored) {
                // keep the original failure
            }
            arena.close();
        }

        /**
         * SimdJsonParser.stage1 (SimdJsonParser.java:55-58) + every StringParser.parseString of the document in ONE native
         * call.  Only buffer[0, length) is read.  On return bitIndexes holds indexes[0..count] (sentinel included) and
         * stringBuffer the records [be32 length][bytes] in structural order.
         */
        void stage1Unescape(byte[] buffer, int length, BitIndexes bitIndexes, byte[] stringBuffer) {
            MemorySegment.copy(buffer, 0, in, JAVA_BYTE, 0, length);
            int rc;
            try {
                rc = (int) STAGE1_UNESCAPE.invokeExact(ctx, in, (long) length, indexes, capacity,
                        out.asSlice(0, 8), out.asSlice(8, 4), strings, capacity,
                        out.asSlice(16, 8), out.asSlice(24, 8), out.asSlice(32, 4));
            } catch (RuntimeException | Error e) {
                throw e;
            } catch (Throwable t) {
                throw new IllegalStateException(t);
            }
            check(rc, "sjmi_stage1_unescape", ctx);
            int count = (int) out.get(JAVA_LONG, 0);
            MemorySegment.copy(indexes, JAVA_INT, 0, bitIndexes.array(), 0, count + 1);
            bitIndexes.setWriteIdx(count);
            throwStage1(out.get(JAVA_INT, 8));
            MemorySegment.copy(strings, JAVA_BYTE, 0, stringBuffer, 0, (int) out.get(JAVA_LONG, 16));
        }

        @Override
        public void close() {
            cleanable.clean();  // (idempotent; the Cleaner runs the same action if the parser is dropped without close())
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
