"""CPU ORACLE for the PLY reader -- TEST INFRASTRUCTURE ONLY (tests/ may import it; the product never does).

A literal restatement of the reference's parser, /root/reference/ply_format/src/ply.ml (cited line by line), so that
path_tracer_ocaml_amd/host/ply.cpp has something to disagree with.  Pure Python: PLY test files are a few KB.

Pinned?  The reference's own tests hold no PLY fixture (ply_format has no test directory) and the real ganesha.ply is
not in the repository (ganesha/README.md:1), so this oracle is pinned by nothing but the source it restates:
PARITY UNPINNED for SURVEY row F2 (also stated in DESIGN.md section 8).

Faithful down to the quirks:
  * Type.int_accessor_exn reads ONE byte for Short / Ushort (ply.ml:104-105) although Type.size says 2 (ply.ml:90);
  * the list parser never advances the input (ply.ml:219-235), so whatever follows a list element is parsed from the
    list element's own first byte;
  * a list element is keyed by its PROPERTY name and holds one column "rows" (ply.ml:234);
  * duplicate element / property names raise (Map.of_alist_exn, ply.ml:194,351);
  * a header without "end_header" never terminates in the reference (Input.read_line returns Some "" for ever,
    ply.ml:36-49 with :289-293): restated here as PlyHang rather than an infinite loop.
"""
import struct


class PlyError(Exception):
    """an Or_error / exception of the reference"""


class PlyHang(PlyError):
    """the reference would loop for ever"""


# Type.t (ply.ml:66-93).  [%of_sexp: t] accepts the constructor name with either capitalisation of its first letter.
_TYPES = ("char", "uchar", "short", "ushort", "int", "uint", "float", "double")
_SIZE = {"char": 1, "uchar": 1, "short": 2, "ushort": 2, "int": 4, "uint": 4, "float": 4, "double": 8}


def type_of_string(s):  # ply.ml:78-86
    if s == "uint8":
        return "uchar"
    if s == "int8":
        return "char"
    low = s[:1].lower() + s[1:]
    if low in _TYPES:
        return low
    raise PlyError(f"unrecognized type {s}")


def float_accessor(ty):  # ply.ml:95-99
    if ty == "float":
        return lambda base, pos: struct.unpack_from("<f", base, pos)[0]  # Int32.float_of_bits (get_int32_le)
    if ty == "double":
        return lambda base, pos: struct.unpack_from("<d", base, pos)[0]
    raise PlyError(f"expected Float|Double, got {ty}")


def int_accessor(ty):  # ply.ml:101-109
    fmt = {"char": "<b", "uchar": "<B", "short": "<b", "ushort": "<B", "int": "<i", "uint": "<I"}.get(ty)  # sic: 1 byte for 16-bit types
    if fmt is None:
        raise PlyError(f"expected integer type, got {ty}")
    return lambda base, pos: struct.unpack_from(fmt, base, pos)[0]


def parse_property(line):  # ply.ml:125-135
    w = line.split(" ")
    if len(w) == 5 and w[0] == "property" and w[1] == "list":
        return ("list", type_of_string(w[2]), type_of_string(w[3]), w[4])
    if len(w) == 3 and w[0] == "property":
        return ("atom", type_of_string(w[1]), w[2])
    raise PlyError(f"cannot parse property: {line}")


def int_of_string(s):  # Base Int.of_string: optional sign, digits with '_' separators, 0x / 0o / 0b prefixes
    t = s.replace("_", "") if (s and s[0] != "_") else "x"
    try:
        sign = -1 if t.startswith("-") else 1
        body = t[1:] if t[:1] in "+-" else t
        if body[:2].lower() in ("0x", "0o", "0b"):
            return sign * int(body, 0)
        if not body.isdigit():
            raise ValueError
        return sign * int(body)
    except ValueError:
        raise PlyError(f"Int.of_string: {s!r}")


def parse_header(buf, pos):  # Header.parse, ply.ml:288-299
    lines = []
    while True:  # loop, ply.ml:289-293
        nl = buf.find(b"\n", pos)
        if nl < 0:
            raise PlyHang('missing "end_header" line: Input.read_line keeps returning Some ""')
        line = buf[pos:nl].decode("latin-1")
        pos = nl + 1
        if line == "end_header":
            break
        lines.append(line)
    fmt_line = next((l for l in lines if l.startswith("format ")), None)  # parse_format, ply.ml:260-267
    if fmt_line is None:
        raise PlyError("header has no format line")
    w = fmt_line.split(" ")
    if not (len(w) == 3 and w[0] == "format" and w[2] == "1.0"):
        raise PlyError(f"cannot parse format line: {fmt_line}")
    fmt = w[1][:1].lower() + w[1][1:]
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise PlyError(f"unrecognized format {w[1]}")
    elements = []  # parse_elements, ply.ml:269-286
    rest = [l for l in lines if l.startswith("element ") or l.startswith("property ")]
    i = 0
    while i < len(rest):
        w = rest[i].split(" ")
        if not (len(w) == 3 and w[0] == "element"):
            raise PlyError(f"expected element: {rest[i]}")
        count = int_of_string(w[2])
        i += 1
        props = []
        while i < len(rest) and rest[i].startswith("property "):
            props.append(parse_property(rest[i]))
            i += 1
        elements.append((w[1], count, props))
    return fmt, elements, pos


def parse_element(elt, buf, pos):  # Element.parse, ply.ml:237-247; returns (key, columns, new_pos)
    name, count, props = elt
    if count < 0:
        raise PlyError("Array.create: negative length")
    if len(props) == 1 and props[0][0] == "list":  # list_parser, ply.ml:219-235
        _, length_type, elt_type, pname = props[0]
        get_len, get_elt = int_accessor(length_type), int_accessor(elt_type)
        len_size, elt_size = _SIZE[length_type], _SIZE[elt_type]
        rows, off = [], pos
        try:
            for _ in range(count):
                n = get_len(buf, off)
                off += len_size
                if n < 0:
                    raise PlyError("Array.init: negative length")
                rows.append([get_elt(buf, off + k * elt_size) for k in range(n)])
                off += n * elt_size
        except struct.error:
            raise PlyError("index out of bounds")  # Bigstring bounds check
        return pname, {"rows": rows}, pos  # sic: the input is NOT advanced
    if all(p[0] == "atom" for p in props):  # fixed_width_parser, ply.ml:208-217 with create_columns :162-195
        width = sum(_SIZE[p[1]] for p in props)
        cols, extract, offset = {}, [], 0
        for _, ty, pname in props:
            if pname in cols:
                raise PlyError(f"Map.of_alist_exn: duplicate key {pname}")
            cols[pname] = (("floats" if ty in ("float", "double") else "ints"), [None] * count)
            extract.append((float_accessor(ty) if ty in ("float", "double") else int_accessor(ty), offset, cols[pname][1]))
            offset += _SIZE[ty]
        try:
            for i in range(count):
                for get, off, col in extract:
                    col[i] = get(buf, pos + width * i + off)
        except struct.error:
            raise PlyError("index out of bounds")
        if pos + width * count > len(buf):
            raise PlyError("Bigsubstring.drop_prefix: beyond the end")  # Input.advance
        return name, cols, pos + width * count
    raise PlyError("TO DO: parse mixed list/non-list element")


def of_bytes(buf):  # of_bigstring, ply.ml:340-352
    if len(buf) < 4:
        raise PlyError("Could not read ply header (not enough bytes)")  # check_file_magic, ply.ml:325-333
    if buf[:4] != b"ply\n":
        raise PlyError('expected file to start with "ply\\n"')
    fmt, elements, pos = parse_header(buf, 4)
    if fmt != "binary_little_endian":
        raise PlyError(f"to do: handle message format {fmt}")
    data = {}
    for elt in elements:  # read_binary_le, ply.ml:335-338
        key, cols, pos = parse_element(elt, buf, pos)
        if key in data:
            raise PlyError(f"Map.of_alist_exn: duplicate key {key}")
        data[key] = cols
    return data


def of_file(path):
    with open(path, "rb") as f:
        return of_bytes(f.read())
