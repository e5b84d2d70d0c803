package org.simdjson;

/**
 * The exceptions StringParser.parseString would have thrown, from the code the GPU string pass leaves in a failing
 * string's record header (FF FF FF code; include/sjmi.h SJMI_E_ESCAPE_UNEXPECTED .. SJMI_E_LOW_SURROGATE_RANGE).
 * Messages: CharacterUtils.java:74-83, StringParser.java:53-55,113-122,127-129.
 */
final class StringErrors {

    static final int ESCAPE_UNEXPECTED = 4;
    static final int INVALID_UNICODE_ESCAPE = 5;
    static final int LOW_SURROGATE_RESERVED = 6;
    static final int LOW_SURROGATE_NO_U = 7;
    static final int LOW_SURROGATE_RANGE = 8;

    /** buffer / idx: the document and the position of the string's opening quote (for the offending character) */
    static JsonParsingException of(int code, byte[] buffer, int idx) {
        switch (code) {
            case ESCAPE_UNEXPECTED:
                return new JsonParsingException("Escaped unexpected character: " + (char) firstBadEscape(buffer, idx));
            case INVALID_UNICODE_ESCAPE:
                return new JsonParsingException("Invalid unicode escape sequence.");
            case LOW_SURROGATE_RESERVED:
                return new JsonParsingException("Invalid code point. The range U+DC00–U+DFFF is reserved for low surrogate.");
            case LOW_SURROGATE_NO_U:
                return new JsonParsingException("Low surrogate should start with '\\u'");
            case LOW_SURROGATE_RANGE:
                return new JsonParsingException("Invalid code point. Low surrogate should be in the range U+DC00–U+DFFF.");
            default:
                return new JsonParsingException("Malformed string (engine code " + code + ")");
        }
    }

    // the character behind the first backslash that is not one of " \ / b f n r t u
    private static byte firstBadEscape(byte[] buffer, int idx) {
        for (int i = idx + 1; i + 1 < buffer.length && buffer[i] != '"'; i++) {
            if (buffer[i] == '\\') {
                byte c = buffer[i + 1];
                if ("\"\\/bfnrtu".indexOf(c) < 0) {
                    return c;
                }
                i++;
            }
        }
        return '?';
    }

    private StringErrors() {
    }
}
