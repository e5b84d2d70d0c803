"""Golden vectors of the on-demand / schema-based front end: inputs, schemas and the values / exception messages the
reference's OWN tests assert (src/test/java/org/simdjson/*SchemaBasedParsingTest.java, cited per block).  The Java schema
classes map to tests/ondemand_common.py schemas as: boolean/Boolean, long/Long (and the narrower integral types where the
asserted message does not depend on the width), double/Double, String, T[] -> ("array", T), RecordWith<T>Field ->
("object", {"field": T}).  Entry = (json text, length or None, schema, expected value or None, expected message or None)."""

F = lambda t: ("object", {"field": t})  # noqa: E731  RecordWith...Field
A = lambda t: ("array", t)  # noqa: E731
MORE = "More than one JSON value at the root of the document, or extra characters at the end of the JSON!"
MINUS = "Invalid number. Minus has to be followed by a digit."
FOLLOWED = "Number has to be followed by a structural character or whitespace."
BOOL2 = "Unrecognized boolean value. Expected: 'true' or 'false'."
BOOL3 = "Unrecognized boolean value. Expected: 'true', 'false' or 'null'."
LONG_RANGE = "Number value is out of long range ([-9223372036854775808, 9223372036854775807])."

VECTORS = []


def add(json, schema, value=None, message=None, length=None):
    VECTORS.append((json, length, schema, value, message))


# ---- BooleanSchemaBasedParsingTest.java ----
for b in (True, False):
    t = "true" if b else "false"
    add(t, "Boolean", b)                                  # :24-36 booleanValueAtRoot
    add(t, "boolean", b)                                  # :38-50
    add('{"field": %s}' % t, F("Boolean"), {"field": b})  # :52-64
    add('{"field": %s}' % t, F("boolean"), {"field": b})  # :66-78
add("null", "Boolean", None)                              # :80-91 nullAtRootWhenBooleanIsExpected
add("null", "boolean", message=BOOL2)                     # :93-105
for j in ('"abc"', "1"):
    add(j, "Boolean", message=BOOL3)                      # :107-120 invalidTypeForBoolean
    add(j, "boolean", message=BOOL2)                      # :122-135
add('{"field": null}', F("Boolean"), {"field": None})     # :137-
add('{"field": null}', F("boolean"), message=BOOL2)

# ---- IntegralNumberSchemaBasedParsingTest.java (long / Long columns) ----
add("null", "Long", None)                                 # :69-81
add("null", "long", message=MINUS)                        # :83-96
add('{"field": null}', F("Long"), {"field": None})        # :124-141
add('{"field": null}', F("long"), message=MINUS)          # :143-161
add("[-128, 1, 127, null]", A("long"), message=MINUS)     # :213-226
add("[-128, 1, 127, null]", A("Long"), [-128, 1, 127, None])
for n in ("9223372036854775808", "9999999999999999999", "10000000000000000000", "-9223372036854775809", "-9999999999999999999",
          "-10000000000000000000"):
    add(n, "long", message=LONG_RANGE)                    # :342-364 outOfPrimitiveLongRange
    add(n, "Long", message=LONG_RANGE)
for t, name, lo, hi in (("byte", "byte", -128, 127), ("short", "short", -32768, 32767), ("int", "int", -2147483648, 2147483647)):
    box = {"byte": "Byte", "short": "Short", "int": "Integer"}[t]
    for n in ("-9223372036854775809", str(lo - 1), str(hi + 1), "9223372036854775808"):
        add(n, t, message="Number value is out of %s range ([%d, %d])." % (name, lo, hi))    # :279-341 outOfPrimitive<T>Range
        add(n, box, message="Number value is out of %s range ([%d, %d])." % (name, lo, hi))
    add(str(lo), t, lo)
    add(str(hi), box, hi)
    add("null", box, None)                                # :69-81
    add("null", t, message=MINUS)                         # :83-96
    add("[%d, 1, %d, null]" % (lo, hi), A(box), [lo, 1, hi, None])
    add("-0", t, 0)                                       # :679-719 minusZeroIsTreatedAs<T>Zero
    add("1.0", t, message=FOLLOWED)                       # :466-489
for n in ("01", "-01", "000", "-000"):
    add(n, "long", message="Invalid number. Leading zeroes are not allowed.")  # :366-389
for n in ("-a123", "--123", "-+123"):
    add(n, "Long", message=MINUS)                         # :391-414
for n in ("-1-2", "1a"):
    add(n, "long", message=FOLLOWED)                      # :416-439
for n in ("123,", "123{}", "1:"):
    add(n, "Long", message=MORE)                          # :441-464
for n in ("1.0", "-1.0", "1e1", "1.9e1"):
    add(n, "long", message=FOLLOWED)                      # :466-489 floatingPointNumberAsIntegralNumber
add("123", "String", message="Invalid value starting at 0. Expected either string or 'null'.")  # :491-510
add("123", "Boolean", message=BOOL3)
add("123", A("long"), message="Expected '[' but got: '1'.")
add('{"field": 123}', F("String"), message="Invalid value starting at 10. Expected either string or 'null'.")  # :512-529
add('{"field": 123}', F("Boolean"), message=BOOL3)
add('{"field": 123}', F(A("long")), message="Expected '[' but got: '1'.")
add("[1, -1, true]", A("long"), message=MINUS)            # :531-553
add('{"field": [1, -1, true]}', F(A("Long")), message=MINUS)  # :555-581
add("[1, -1, 0]", "String", message="Invalid value starting at 0. Expected either string or 'null'.")  # :583-605
add("[1, -1, 0]", "long", message=MINUS)
add("[1, -1, 0]", A(A("long")), message="Expected '[' but got: '1'.")
add('{"field": [1, -1, 0]}', A("long"), message="Expected '[' but got: '{'.")  # :607-628
add('{"field": [1, -1, 0]}', F("Boolean"), message=BOOL3)
add('{"field": [1, -1, 0]}', F("boolean"), message=BOOL2)
add('{"field": [1, -1, 0]}', F("String"), message="Invalid value starting at 10. Expected either string or 'null'.")
add('{"field": [1, -1, 0]}', "long", message=MINUS)
add("+1", "long", message=MINUS)                          # :630-652
for n in ("a123", "a-123"):
    add(n, "Long", message=MINUS)                         # :654-677
add("-0", "long", 0)                                      # :721-733
add("", "Long", message="No structural element found.")  # :735-748
add("null", "Long", message="Invalid value starting at 0. Expected 'null'.", length=3)  # :750-763
add("1234", "long", 12, length=2)                         # :765-777
add("9223372036854775807", "long", 9223372036854775807)
add("-9223372036854775808", "Long", -9223372036854775808)

# ---- FloatingPointNumberSchemaBasedParsingTest.java (double / Double columns) ----
add("null", "Double", None)                               # :54-66
add("null", "double", message=MINUS)                      # :68-80
add('{"field": null}', F("double"), message=MINUS)        # :119-131
add("[-1.1, 1.0, 0.0, null]", A("double"), message=MINUS)  # :164-176
add("[-1.1, 1.0, 0.0, null]", A("Double"), [-1.1, 1.0, 0.0, None])
for n in ("01.0", "-01.0", "000.0", "-000.0", "012e34"):
    add(n, "double", message="Invalid number. Leading zeroes are not allowed.")  # :222-236
for n in ("-a123.0", "--123.0", "-+123.0", "-.123", "-e123"):
    add(n, "Double", message=MINUS)                       # :238-252
for n in ("-1.0-2", "1.0a", "12E12.12", "1e2e3"):
    add(n, "double", message=FOLLOWED)                    # :254-268
for n in ("123.", "1..1", "1.e1", "1.E1"):
    add(n, "double", message="Invalid number. Decimal point has to be followed by a digit.")  # :270-284
for n in ("1e+-2", "1E+-2", "1e--23", "1E--23", "1ea", "1Ea", "1e", "1E", "1e+", "1E+"):
    add(n, "Double", message="Invalid number. Exponent indicator has to be followed by a digit.")  # :286-300
add("+1.0", "double", message=MINUS)                      # :302-315
for n in ("a123", "a-123"):
    add(n, "double", message=MINUS)                       # :317-331
for n in ("123.0,", "123.0{}", "1.0:"):
    add(n, "double", message=MORE)                        # :1011-1025
add("123.0", "String", message="Invalid value starting at 0. Expected either string or 'null'.")  # :1027-1046
add("123.0", "Boolean", message=BOOL3)
add("123.0", A("long"), message="Expected '[' but got: '1'.")
add('{"field": 123.0}', F("String"), message="Invalid value starting at 10. Expected either string or 'null'.")  # :1048-1065
add("[1.0, -1.0, true]", A("double"), message=MINUS)      # :1067-1080
add("1.5", "double", 1.5)
add("-2.5e3", "Double", -2500.0)
add("1e400", "double", float("inf"))
add("-1e400", "double", float("-inf"))
add("1e-400", "double", 0.0)
add("123", "double", message="Invalid floating-point number. Fraction or exponent part is missing.")  # NumberParser.java:302-304

# ---- StringSchemaBasedParsingTest.java ----
add('""', "String", b"")                                  # :34-45
add('"abc"', "String", b"abc")
for j in ("true", "false", "1"):
    add(j, "String", message="Invalid value starting at 0. Expected either string or 'null'.")   # :61-74
    add('{"field": %s}' % j, F("String"), message="Invalid value starting at 10. Expected either string or 'null'.")  # :165-181
add("null", "String", None)                               # :76-87
for j in ('"abc",', '"abc"def'):
    add(j, "String", message=MORE)                        # :110-123
add('{"field": ""}', F("String"), {"field": b""})         # :125-136
add('{"field": null}', F("String"), {"field": None})      # :152-163
add('"\\uDC00"', "String", message="Invalid code point. The range U+DC00–U+DFFF is reserved for low surrogate.")  # :224-240
add('"a\\nb\\u00e9\\uD83D\\uDE00"', "String", "a\nbé\U0001F600".encode())
add('"abc"', "Boolean", message=BOOL3)                    # :89-108 mismatchedTypeForStringAsRoot
add('"abc"', "long", message=MINUS)

for ch, want in (("a", 0x61), ("\\n", 10), ("\\u0041", 0x41), ("é", 0xE9), ("€", 0x20AC), ("\\\\", 0x5C), ("\\\"", 0x22)):
    add('"%s"' % ch, "Character", want)                   # :244-272 characterAtRoot / primitiveCharAtRoot (shape)
    add('"%s"' % ch, "char", want)
    add('{"field": "%s"}' % ch, F("char"), {"field": want})  # :334-366
add("null", "Character", None)                            # :276-287
add("null", "char", message="Invalid value starting at 0. Expected string.")  # :289-301
for j in ("true", "false", "1"):
    add(j, "Character", message="Invalid value starting at 0. Expected either string or 'null'.")  # :303-316
    add(j, "char", message="Invalid value starting at 0. Expected string.")                       # :318-331
    add('{"field": %s}' % j, F("Character"), message="Invalid value starting at 10. Expected either string or 'null'.")  # :394-409
    add('{"field": %s}' % j, F("char"), message="Invalid value starting at 10. Expected string.")                        # :412-427
add('{"field": null}', F("Character"), {"field": None})   # :349-360
add('{"field": null}', F("char"), message="Invalid value starting at 10. Expected string.")  # :377-392
add("a", "Character", message="Invalid value starting at 0. Expected either string or 'null'.")  # :460-471
add('"ab"', "char", message="String cannot be deserialized to a char. Expected a single-character string.")
add('"\\uD83D\\uDE00"', "char", message="Invalid code point. Should be within the range U+0000–U+D777 or U+E000–U+FFFF.")
add('"😀"', "Character", message="String cannot be deserialized to a char. Expected a single 16-bit code unit character.")
add('"\\u12G4"', "char", message="Invalid unicode escape sequence.")

# ---- ArraySchemaBasedParsingTest.java ----
add("[]", A("long"), [])                                  # :50-63
add('{"field": []}', F(A("Long")), {"field": []})         # :65-76
for j in ("1", "true", "false", "{}", ":", ",", '"abc"'):
    add(j, A("long"), message="Expected '[' but got: '%s'." % j[0])  # :78-91
add("[1 1]", A("long"), message="Missing comma between array values")  # :93-105
for j in ("[1,,1]", "[,]", "[,,]"):
    add(j, A("long"), message=MINUS)                      # :107-120
    add('{"field": %s}' % j, F(A("long")), message=MINUS)  # :212-228
for j in ("[,", "[1 ", "[,,", "[1,", "[1", "["):
    add(j, A("long"), message="Unclosed array. Missing ']' for starting '['.")  # :122-135
add("[[]]", A(A("long")), message="Missing comma between array values", length=3)  # :137-149
add("[[[[", A("long"), message="Unclosed array. Missing ']' for starting '['.", length=2)  # :151-163
add("[][[[[", A("long"), [], length=2)                    # :165-176
add('{"field": [1 1]}', F(A("long")), message="Missing comma between array values")  # :178-193
add("null", A("long"), None)                              # :390-401
add('{"field": null}', F(A("long")), {"field": None})     # :403-415
add("", A("long"), message="No structural element found.")  # :467-479
add("null", A("Boolean"), message="Invalid value starting at 0. Expected 'null'.", length=3)  # :481-493
add("[[1,2],[3],[]]", A(A("long")), [[1, 2], [3], []])    # :333-345 multidimensionalArrays2d (shape)

# ---- ObjectSchemaBasedParsingTest.java ----
add("{}", F("Long"), {"field": None})                     # :45-75
add("null", F("Long"), None)                              # :91-102
add('{"nestedField": null}', ("object", {"nestedField": F("String")}), {"nestedField": None})  # :104-116
add('{"\\"abc\\\\": 1}', ("object", {'"abc\\': "long"}), {'"abc\\': 1})  # :198-209 fieldNamesWithEscapes
add('{"first": 1, "field": 2, "second": 3}', F("long"), {"field": 2})   # :211-222
add('{"first": 1, "second": 3}', F("Long"), {"field": None})            # :224-235
add('{"nestedField": {}}', ("object", {"nestedField": F("String")}), {"nestedField": {"field": None}})  # :255-268
add('{"nestedField": {"field": "abc"}}', ("object", {"nestedField": F("String")}), {"nestedField": {"field": b"abc"}})  # :270-283
add('"{}"', F("String"), message="Expected '{' but got: '\"'.")         # :285-300
add('{"nestedField": true}', ("object", {"nestedField": F("String")}), message="Expected '{' but got: 't'.")  # :302-317
add('{: 2, "field": 1}', F("long"), message="Expected '\"' but got: ':'.")  # :395-410
for name in ("\\null", "1", "true", "false", "[]", "{}"):
    add("{" + name + ": 1}", F("long"), message="Expected '\"' but got: '%s'." % name[0])  # :412-428
for j in ('{"field": 1', '{"field":', '{"field"', "{", '{"ignore": {"field": 1', '{"field": 1,'):
    add(j, F("long"), message="Unclosed object. Missing '}' for starting '{'.")  # :430-445
add("", F("long"), message="No structural element found.")  # :576-591
add("null", F("long"), message="Invalid value starting at 0. Expected 'null'.", length=3)  # :593-608
add('{"name": "John", "age": 30, "aaa": 1, "bbb": 2, "ccc": 3}', ("object", {"name": "String", "age": "long"}),
    {"name": b"John", "age": 30})                         # :644- issue50
add('[{"field": 1}, {"field": 2}, {}, null]', A(F("Long")), [{"field": 1}, {"field": 2}, {"field": None}, None])  # :447-459 (shape)
