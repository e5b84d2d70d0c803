"""simdjson-java_amd -- MI355X-native stage-1 engine behind simdjson-java's SimdJsonParser.parse.

Only what the hot path needs lives here:
  csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/sjmi.h) -> libsjmi.so
  binding.py ctypes view of that C ABI for the Python test/bench harness
There is deliberately NO CPU fallback: without libsjmi.so + a GPU every call raises.
"""
from .binding import (Context, SjmiError, build, lib, lib_path, ST_CAPACITY, ST_INTERNAL, ST_UNCLOSED,  # noqa: F401
                      ST_UNESCAPED, ST_UTF8, PADDING, status_message, SimdJsonParser, JsonParsingException,
                      ParsedDocument, JsonValue)
