"""Shared by the CPU and GPU tests of the on-demand front end (csrc/host/ondemand.h): a schema driver that makes the calls
SchemaBasedJsonIterator makes (SchemaBasedJsonIterator.java:29-132,:229-272,:481-502,:720-734) on ANY iterator with the
method names of oracle/ondemand.py, and a schema-less fuzz driver that walks a document by peeking (with seeded random
skips, wrong-typed reads and early exits) and records every result and exception message as a trace.

Schemas: "boolean" "byte" "short" "int" "long" "double" (primitive: the NonNull getters), "Boolean" "Byte" "Short" "Integer" "Long"
"Double" "String" (nullable),
("array", element schema), ("object", {field name: schema})."""
from oracle import ondemand as OD

PRIMITIVE = {"boolean", "byte", "short", "int", "long", "float", "double"}
INTEGRAL = {"byte": 8, "Byte": 8, "short": 16, "Short": 16, "int": 32, "Integer": 32}


def _scalar(it, schema, root):
    if schema in ("boolean", "Boolean"):
        return it.get_boolean(root=root, nullable=schema == "Boolean")
    if schema in ("long", "Long"):
        return it.get_long(root=root, nullable=schema == "Long")
    if schema in INTEGRAL:
        return it.get_long(root=root, nullable=schema[0].isupper(), bits=INTEGRAL[schema])
    if schema in ("double", "Double"):
        return it.get_double(root=root, nullable=schema == "Double")
    if schema in ("float", "Float"):
        return it.get_float(root=root, nullable=schema == "Float")
    if schema in ("char", "Character"):
        return it.get_char(root=root, nullable=schema == "Character")
    if schema == "String":
        return it.get_string(root=root)
    raise ValueError(schema)


def _object(it, fields, result):  # getObject :68-86
    if result == OD.NOT_EMPTY:
        args = {k: None for k in fields}
        parent_depth = it.depth_value() - 1
        collected, has_fields = 0, True
        while collected < len(fields) and has_fields:  # collectArguments :96-113
            name = it.get_field_name()
            it.move_to_field_value()
            key = name.decode("utf-8", "replace")
            if key in fields:
                args[key] = _value(it, fields[key])
                collected += 1
            else:
                it.skip_child()
            has_fields = it.next_object_field()
        it.skip_child(parent_depth)
        return args
    if result == OD.EMPTY:
        return {k: None for k in fields}
    return None


def _array(it, elem, result):  # getArray :241-272
    if result == OD.EMPTY:
        return []
    if result == OD.NULL:
        return None
    out, has = [], True
    while has:
        out.append(_value(it, elem))
        has = it.next_array_element()
    return out


def _value(it, schema):  # collectArgument :115-132 (non-root forms)
    if isinstance(schema, tuple) and schema[0] == "array":
        return _array(it, schema[1], it.start_iterating_array(root=False))
    if isinstance(schema, tuple) and schema[0] == "object":
        return _object(it, schema[1], it.start_iterating_object(root=False))
    return _scalar(it, schema, False)


def walk_document(it, schema):
    """SchemaBasedJsonIterator.walkDocument :29-59 after jsonIterator.init"""
    if isinstance(schema, tuple) and schema[0] == "array":  # getRootArray :229-234
        v = _array(it, schema[1], it.start_iterating_array(root=True))
        it.assert_no_more_json_values()
        return v
    if isinstance(schema, tuple) and schema[0] == "object":  # getRootObject :61-66
        v = _object(it, schema[1], it.start_iterating_object(root=True))
        it.assert_no_more_json_values()
        return v
    return _scalar(it, schema, True)


class OracleIterator(OD.OnDemandJsonIterator):
    def depth_value(self):
        return self.depth


def run_oracle(doc, length, indexes, schema):
    """-> ("ok", value) or ("error", message)"""
    try:
        return "ok", walk_document(OracleIterator(doc, length, indexes), schema)
    except OD.JsonParsingException as e:
        return "error", str(e)


# ---- schema-less fuzz driver: the same seeded decisions on both implementations -> comparable traces ----
def fuzz_walk(it, rng, trace, root=True, budget=None):
    """Walk the value at the cursor by its first byte; sometimes skip it, read it with the wrong getter, or leave a container
    early with skipChild(parentDepth).  Every result goes to `trace`; an exception propagates (the caller records it)."""
    b = it.peek_byte()
    r = rng.random()
    if not root and r < 0.12:
        it.skip_child()
        trace.append(("skip", it.depth_value()))
        return
    wrong = r > 0.97
    if b == 0x5B or (wrong and rng.random() < 0.2):
        res = it.start_iterating_array(root=root)
        trace.append(("array", res))
        if res == OD.NOT_EMPTY:
            parent = it.depth_value() - 2
            has = True
            while has:
                fuzz_walk(it, rng, trace, False)
                if rng.random() < 0.05:
                    it.skip_child(parent)
                    trace.append(("leave-array", it.depth_value()))
                    break
                has = it.next_array_element()
                trace.append(("next", has))
        if root:
            it.assert_no_more_json_values()
    elif b == 0x7B or (wrong and rng.random() < 0.3):
        res = it.start_iterating_object(root=root)
        trace.append(("object", res))
        if res == OD.NOT_EMPTY:
            parent = it.depth_value() - 1
            has = True
            while has:
                name = it.get_field_name()
                it.move_to_field_value()
                trace.append(("field", name))
                fuzz_walk(it, rng, trace, False)
                if rng.random() < 0.05:
                    it.skip_child(parent)
                    trace.append(("leave-object", it.depth_value()))
                    break
                has = it.next_object_field()
                trace.append(("next", has))
            else:
                it.skip_child(parent)
        if root:
            it.assert_no_more_json_values()
    elif b == 0x22 and not wrong:
        if rng.random() < 0.2:
            trace.append(("char", it.get_char(root=root, nullable=rng.random() < 0.5)))
        else:
            trace.append(("string", it.get_string(root=root)))
    elif b in b"tf" and not wrong:
        trace.append(("boolean", it.get_boolean(root=root, nullable=rng.random() < 0.5)))
    elif b == 0x6E and not wrong:
        k = rng.randrange(4)
        trace.append(("null", [it.get_boolean, it.get_long, it.get_double][k](root=root, nullable=True) if k < 3 else it.get_string(root=root)))
    else:
        k = rng.randrange(5 if wrong else 2)
        if k == 1 and rng.random() < 0.3:
            v = it.get_float(root=root, nullable=rng.random() < 0.5)
            trace.append(("float", None if v is None else OD.float_bits(v)))
        elif k == 0:
            trace.append(("long", it.get_long(root=root, nullable=rng.random() < 0.5)))
        elif k == 1:
            v = it.get_double(root=root, nullable=rng.random() < 0.5)
            trace.append(("double", None if v is None else OD.double_bits(v)))
        elif k == 2:
            trace.append(("boolean", it.get_boolean(root=root, nullable=True)))
        elif k == 3:
            trace.append(("string", it.get_string(root=root)))
        else:
            it.skip_child()
            trace.append(("skip", it.depth_value()))
    trace.append(("depth", it.depth_value()))
