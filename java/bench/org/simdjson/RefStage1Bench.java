package org.simdjson;

import java.nio.file.Files;
import java.nio.file.Paths;
import java.util.Locale;
import java.util.concurrent.CountDownLatch;
import java.util.concurrent.atomic.AtomicLong;

/**
 * Timing harness for the REFERENCE's own CPU path, beside the GPU figures of bench.py (BASELINE.md 4, SURVEY.md 8(d)).
 *
 * Not part of the product and not part of the reference: it lives in package org.simdjson only because the two stage-1 passes
 * it times are package-private there --
 *     Utf8Validator.validate(byte[], int)      (Utf8Validator.java:54)
 *     StructuralIndexer.index(byte[], int)     (StructuralIndexer.java:38, :196-303 at 512 bits)
 * i.e. exactly the body of SimdJsonParser.stage1 (SimdJsonParser.java:55-58); the second measurement is the public
 * SimdJsonParser.parse(byte[], int) (SimdJsonParser.java:35-40), the shape of jmh/ParseBenchmark.java:40-48.
 *
 * Build + run (bench.py does this when a JDK >= 24 and the reference's sources are on the box):
 *     javac --add-modules jdk.incubator.vector -d OUT $(find REF/src/main/java -name '*.java') java/bench/org/simdjson/RefStage1Bench.java
 *     java  --add-modules jdk.incubator.vector -Dorg.simdjson.species=512 -cp OUT org.simdjson.RefStage1Bench FILE THREADS SECONDS [SLICE_BYTES]
 *
 * One parser (one BitIndexes + one StructuralIndexer) per thread, as the reference intends (SimdJsonParser is not thread-safe);
 * stage 1 runs over a slice of at most SLICE_BYTES (default 32 MiB: below the parser's default 34 MiB capacity) made of whole
 * copies of FILE; T is 1 and then THREADS.  Prints ONE JSON line.
 */
public final class RefStage1Bench {

    private static final int PADDING = 64;  // SimdJsonParser.java:5

    private RefStage1Bench() {
    }

    private interface Work {
        /** one pass; returns the bytes (stage 1) or documents (parse) it processed */
        long pass();
    }

    private static double run(int threads, double seconds, java.util.function.IntFunction<Work> make) throws InterruptedException {
        final Work[] works = new Work[threads];
        for (int t = 0; t < threads; t++) {
            works[t] = make.apply(t);
            works[t].pass();  // warm-up: class loading, C2 compilation of the vector loops
            works[t].pass();
        }
        final AtomicLong units = new AtomicLong();
        final CountDownLatch ready = new CountDownLatch(threads);
        final CountDownLatch go = new CountDownLatch(1);
        final long[] span = new long[2];
        Thread[] th = new Thread[threads];
        for (int t = 0; t < threads; t++) {
            final Work w = works[t];
            th[t] = new Thread(() -> {
                ready.countDown();
                try {
                    go.await();
                } catch (InterruptedException e) {
                    return;
                }
                final long stop = span[0] + (long) (seconds * 1e9);
                long mine = 0;
                while (System.nanoTime() < stop) {
                    mine += w.pass();
                }
                units.addAndGet(mine);
            });
            th[t].start();
        }
        ready.await();
        span[0] = System.nanoTime();
        go.countDown();
        for (Thread x : th) {
            x.join();
        }
        span[1] = System.nanoTime();
        return units.get() / ((span[1] - span[0]) / 1e9);
    }

    public static void main(String[] args) throws Exception {
        if (args.length < 3) {
            System.err.println("usage: RefStage1Bench FILE THREADS SECONDS [SLICE_BYTES]");
            System.exit(2);
        }
        final byte[] doc = Files.readAllBytes(Paths.get(args[0]));
        final int threads = Integer.parseInt(args[1]);
        final double seconds = Double.parseDouble(args[2]);
        final long sliceMax = args.length > 3 ? Long.parseLong(args[3]) : 32L << 20;
        final int copies = (int) Math.max(1, Math.min(sliceMax, Integer.MAX_VALUE - 2L * PADDING) / doc.length);
        final int len = copies * doc.length;
        final byte[] slice = new byte[len + PADDING];  // (padded: padIfNeeded copies nothing, SimdJsonParser.java:42-48)
        for (int k = 0; k < copies; k++) {
            System.arraycopy(doc, 0, slice, k * doc.length, doc.length);
        }
        // one stage-1 pass up front for the structural count of the slice (BitIndexes: writeIdx is not exposed; count by cursor)
        final BitIndexes probe = new BitIndexes(len + PADDING);
        final StructuralIndexer probeIndexer = new StructuralIndexer(probe);
        Utf8Validator.validate(slice, len);
        probeIndexer.index(slice, len);
        long structurals = 0;
        while (probe.hasNext()) {
            probe.getAndAdvance();
            structurals++;
        }
        final java.util.function.IntFunction<Work> stage1 = t -> {
            final byte[] mine = t == 0 ? slice : slice.clone();  // (every thread streams its own copy, like independent parsers would)
            final BitIndexes bits = new BitIndexes(len + PADDING);
            final StructuralIndexer indexer = new StructuralIndexer(bits);
            return () -> {
                bits.reset();                           // SimdJsonParser.reset, :50-53
                Utf8Validator.validate(mine, len);      // SimdJsonParser.stage1, :55-58
                indexer.index(mine, len);
                return len;
            };
        };
        final byte[] padded = java.util.Arrays.copyOf(doc, doc.length + PADDING);
        final java.util.function.IntFunction<Work> parse = t -> {
            final byte[] mine = padded.clone();
            final SimdJsonParser parser = new SimdJsonParser();
            return () -> {
                JsonValue v = parser.parse(mine, doc.length);  // SimdJsonParser.java:35-40
                return v == null ? 0 : 1;
            };
        };
        final double s1One = run(1, seconds, stage1);
        final double s1All = threads > 1 ? run(threads, seconds, stage1) : s1One;
        final double pOne = run(1, Math.min(seconds, 3.0), parse);
        final double pAll = threads > 1 ? run(threads, Math.min(seconds, 3.0), parse) : pOne;
        System.out.println(String.format(Locale.ROOT,
                "{\"harness\":\"RefStage1Bench\",\"species\":\"%s\",\"vector_bits\":%d,\"java\":\"%s\",\"threads\":%d,"
                        + "\"slice_bytes\":%d,\"slice_structurals\":%d,\"seconds\":%.2f,"
                        + "\"stage1_gb_per_s_one_thread\":%.4f,\"stage1_gb_per_s_all_threads\":%.4f,"
                        + "\"parse_per_s_one_thread\":%.2f,\"parse_per_s_all_threads\":%.2f}",
                System.getProperty("org.simdjson.species", "preferred"), VectorUtils.BYTE_SPECIES.vectorBitSize(),
                System.getProperty("java.version"), threads, len, structurals, seconds,
                s1One / 1e9, s1All / 1e9, pOne, pAll));
    }
}
