// User-side code of the reference's headline "parse and select" benchmark (jmh ParseAndSelectBenchmark /
// SchemaBasedParseAndSelectBenchmark, BenchmarkCorrectnessTest.java:23-55: the screen names of twitter.json's users with
// default_profile) written against the PUBLIC C ABI (include/sjmi.h), the way a host-language binding would call it:
//   mode 0  full parse (sjmi_parser_parse) + JsonValue walk (sjmi_value_*)
//   mode 1  on-demand cursor (sjmi_parser_ondemand_init + sjmi_od_*), skipChild scanning like the reference
//   mode 3  sjmi_parser_parse alone (any document): the binding-free cost of SimdJsonParser.parse
// Built and loaded by bench.py (section `select`) / tools/ondemand_bench.py; links libsjmi.so.
#include <stdint.h>
#include <string.h>

#include <chrono>

#include "../include/sjmi.h"

namespace {
bool is(const uint8_t* p, uint64_t n, const char* s) { return n == strlen(s) && memcmp(p, s, n) == 0; }

// SchemaBasedJsonIterator.getObject / collectArguments (:68-113) for {default_profile: boolean, screen_name: String}
int select_user(sjmi_parser* p, int* is_default, uint64_t* name_len) {
    int res = 0, rc;
    if ((rc = sjmi_od_start_object(p, 0, &res))) return rc;
    if (res != SJMI_OD_NOT_EMPTY) return 0;
    const int parent = sjmi_od_depth(p) - 1;
    int collected = 0, has = 1;
    while (collected < 2 && has) {
        const uint8_t* name;
        uint64_t n;
        if ((rc = sjmi_od_get_field_name(p, &name, &n))) return rc;
        const bool dp = is(name, n, "default_profile"), sn = is(name, n, "screen_name");
        if ((rc = sjmi_od_move_to_field_value(p))) return rc;
        if (dp) {
            int isnull;
            if ((rc = sjmi_od_get_boolean(p, 0, 0, &isnull, is_default))) return rc;
            ++collected;
        } else if (sn) {
            int isnull;
            const uint8_t* s;
            if ((rc = sjmi_od_get_string(p, 0, &isnull, &s, name_len))) return rc;
            ++collected;
        } else if ((rc = sjmi_od_skip_child(p, -1))) {
            return rc;
        }
        if ((rc = sjmi_od_next_object_field(p, &has))) return rc;
    }
    return sjmi_od_skip_child(p, parent);
}

int select_status(sjmi_parser* p, uint64_t* selected, uint64_t* bytes) {
    int res = 0, rc;
    if ((rc = sjmi_od_start_object(p, 0, &res))) return rc;
    if (res != SJMI_OD_NOT_EMPTY) return 0;
    const int parent = sjmi_od_depth(p) - 1;
    int collected = 0, has = 1;
    while (collected < 1 && has) {
        const uint8_t* name;
        uint64_t n;
        if ((rc = sjmi_od_get_field_name(p, &name, &n))) return rc;
        const bool user = is(name, n, "user");
        if ((rc = sjmi_od_move_to_field_value(p))) return rc;
        if (user) {
            int dflt = 0;
            uint64_t len = 0;
            if ((rc = select_user(p, &dflt, &len))) return rc;
            if (dflt) {
                ++*selected;
                *bytes += len;
            }
            ++collected;
        } else if ((rc = sjmi_od_skip_child(p, -1))) {
            return rc;
        }
        if ((rc = sjmi_od_next_object_field(p, &has))) return rc;
    }
    return sjmi_od_skip_child(p, parent);
}

int select_on_demand(sjmi_parser* p, const uint8_t* buf, uint64_t len, uint64_t* selected, uint64_t* bytes) {
    int rc, res = 0;
    if ((rc = sjmi_parser_ondemand_init(p, buf, len, 0))) return rc;
    if ((rc = sjmi_od_start_object(p, 1, &res))) return rc;
    if (res != SJMI_OD_NOT_EMPTY) return 0;
    const int parent = sjmi_od_depth(p) - 1;
    int collected = 0, has = 1;
    while (collected < 1 && has) {
        const uint8_t* name;
        uint64_t n;
        if ((rc = sjmi_od_get_field_name(p, &name, &n))) return rc;
        const bool statuses = is(name, n, "statuses");
        if ((rc = sjmi_od_move_to_field_value(p))) return rc;
        if (statuses) {
            if ((rc = sjmi_od_start_array(p, 0, &res))) return rc;
            int more = res == SJMI_OD_NOT_EMPTY;
            while (more) {
                if ((rc = select_status(p, selected, bytes))) return rc;
                if ((rc = sjmi_od_next_array_element(p, &more))) return rc;
            }
            ++collected;
        } else if ((rc = sjmi_od_skip_child(p, -1))) {
            return rc;
        }
        if ((rc = sjmi_od_next_object_field(p, &has))) return rc;
    }
    if ((rc = sjmi_od_skip_child(p, parent))) return rc;
    return sjmi_od_assert_no_more_values(p);
}

int select_full_parse(sjmi_parser* p, const uint8_t* buf, uint64_t len, uint64_t* selected, uint64_t* bytes) {
    const uint64_t* tape;
    const uint8_t* strings;
    uint64_t tl, sl, pos;
    int rc;
    if ((rc = sjmi_parser_parse(p, buf, len, &tape, &tl, &strings, &sl, &pos))) return rc;
    sjmi_value root, statuses, st, user, v;
    if (sjmi_parser_root(p, &root) || sjmi_value_get(p, &root, (const uint8_t*)"statuses", 8, &statuses)) return -100;
    for (int more = sjmi_value_first(p, &statuses, &st); more == 0; more = sjmi_value_next(p, &statuses, &st, &st)) {
        if (sjmi_value_get(p, &st, (const uint8_t*)"user", 4, &user)) return -101;
        int dflt = 0;
        if (sjmi_value_get(p, &user, (const uint8_t*)"default_profile", 15, &v) || sjmi_value_as_boolean(p, &v, &dflt)) return -102;
        if (dflt) {
            uint8_t name[256];
            uint64_t n = 0;
            if (sjmi_value_get(p, &user, (const uint8_t*)"screen_name", 11, &v) || sjmi_value_as_string(p, &v, name, sizeof name, &n)) return -103;
            ++*selected;
            *bytes += n;
        }
    }
    return 0;
}
}  // namespace

// BASELINE.json configs[4]: a batch of documents from a host buffer -> one tree per document (sjmi_parser_parse_batch: GPU
// stage 1 + GPU string records + the host stage 2 on a thread pool), timed here; then, untimed, BenchmarkCorrectnessTest's
// selection (:23-55) on EVERY tree through sjmi_value_*: *users_min / *users_max over the documents.
extern "C" int odb_parse_batch(sjmi_parser* p, const uint8_t* buf, uint64_t total_len, const uint64_t* offsets, uint64_t n_docs, int iters,
                               double* seconds, uint64_t* ok_docs, uint64_t* users_min, uint64_t* users_max) {
    const uint64_t *tape, *tape_offsets;
    const uint8_t* strings;
    const int32_t* errors;
    uint64_t sl = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) {
        const int rc = sjmi_parser_parse_batch(p, buf, total_len, offsets, n_docs, &tape, &tape_offsets, &strings, &sl, &errors);
        if (rc) return rc;
    }
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *ok_docs = 0;
    *users_min = ~0ull;
    *users_max = 0;
    for (uint64_t k = 0; k < n_docs; ++k) {
        if (errors[k]) continue;
        ++*ok_docs;
        sjmi_value root, statuses, st, user, v;
        if (sjmi_parser_batch_root(p, k, &root) || sjmi_value_get(p, &root, (const uint8_t*)"statuses", 8, &statuses)) return -100;
        uint64_t users = 0;
        for (int more = sjmi_value_first(p, &statuses, &st); more == 0; more = sjmi_value_next(p, &statuses, &st, &st)) {
            int dflt = 0;
            if (sjmi_value_get(p, &st, (const uint8_t*)"user", 4, &user) ||
                sjmi_value_get(p, &user, (const uint8_t*)"default_profile", 15, &v) || sjmi_value_as_boolean(p, &v, &dflt))
                return -101;
            users += dflt != 0;
        }
        if (users < *users_min) *users_min = users;
        if (users > *users_max) *users_max = users;
    }
    return 0;
}

extern "C" int odb_run(sjmi_parser* p, const uint8_t* buf, uint64_t len, int mode, int iters, double* seconds, uint64_t* selected,
                       uint64_t* bytes) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) {
        *selected = 0;
        *bytes = 0;
        int rc;
        if (mode == 3) {  // the parse alone (any document)
            const uint64_t* tape;
            const uint8_t* strings;
            uint64_t tl, sl, pos;
            rc = sjmi_parser_parse(p, buf, len, &tape, &tl, &strings, &sl, &pos);
            *selected = tl;
            *bytes = sl;
        } else {
            rc = mode == 0 ? select_full_parse(p, buf, len, selected, bytes) : select_on_demand(p, buf, len, selected, bytes);
        }
        if (rc) return rc;
    }
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}
