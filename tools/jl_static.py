"""Static, execute-nothing reader for the reference's joblib ``.jl`` fixtures.

The reference's test inputs ``repo_utils/test_files/chunk{0,1,2}.jl`` are joblib
pickles (written by ``utmos/convert.py:98``).  Unpickling runs code chosen by the
file, so this tool never unpickles: it zlib-inflates the container and walks the
pickle opcode stream with ``pickletools.genops`` (a disassembler), rebuilding only
plain data -- str / int / float / bool / None / tuple / list / dict -- and recording
every GLOBAL / REDUCE / NEWOBJ / BUILD as an inert ``Sym`` record.  The raw array
bytes joblib appends after each ``NumpyArrayWrapper`` are sliced out by shape and
dtype string.  Nothing from the file is imported, called or instantiated.

Only used in the build container to re-encode fixtures (tools/make_golden.py);
never at test time or on the GPU box.
"""
import io
import pickletools
import zlib

import numpy as np


class Sym:
    """Inert record of a pickle GLOBAL/REDUCE/NEWOBJ (never resolved or called)."""

    def __init__(self, kind, *parts):
        self.kind = kind
        self.parts = parts
        self.state = None

    def __repr__(self):
        return f"Sym({self.kind}, {self.parts!r}, state={self.state!r})"


_MARK = object()


class _Machine:
    def __init__(self, fh):
        self.fh = fh
        self.stack = []
        self.memo = {}
        self.memo_next = 0

    def pop_mark(self):
        items = []
        while True:
            x = self.stack.pop()
            if x is _MARK:
                break
            items.append(x)
        items.reverse()
        return items

    def run(self):
        """Interpret opcodes until STOP; returns the top of stack."""
        st = self.stack
        for op, arg, _pos in pickletools.genops(self.fh):
            n = op.name
            if n in ("PROTO", "FRAME"):
                continue
            if n == "STOP":
                return st.pop()
            if n == "MARK":
                st.append(_MARK)
            elif n in ("SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8", "UNICODE",
                       "BININT", "BININT1", "BININT2", "LONG1", "LONG4", "INT", "LONG",
                       "BINFLOAT", "FLOAT", "SHORT_BINBYTES", "BINBYTES", "BINBYTES8",
                       "SHORT_BINSTRING", "BINSTRING", "STRING"):
                st.append(arg)
            elif n == "NONE":
                st.append(None)
            elif n == "NEWTRUE":
                st.append(True)
            elif n == "NEWFALSE":
                st.append(False)
            elif n == "EMPTY_DICT":
                st.append({})
            elif n == "EMPTY_LIST":
                st.append([])
            elif n == "EMPTY_TUPLE":
                st.append(())
            elif n == "TUPLE1":
                a = st.pop(); st.append((a,))
            elif n == "TUPLE2":
                b = st.pop(); a = st.pop(); st.append((a, b))
            elif n == "TUPLE3":
                c = st.pop(); b = st.pop(); a = st.pop(); st.append((a, b, c))
            elif n == "TUPLE":
                st.append(tuple(self.pop_mark()))
            elif n == "LIST":
                st.append(list(self.pop_mark()))
            elif n == "DICT":
                it = self.pop_mark(); st.append(dict(zip(it[0::2], it[1::2])))
            elif n == "APPEND":
                v = st.pop(); st[-1].append(v)
            elif n == "APPENDS":
                it = self.pop_mark(); st[-1].extend(it)
            elif n == "SETITEM":
                v = st.pop(); k = st.pop(); self._setitem(st[-1], k, v)
            elif n == "SETITEMS":
                it = self.pop_mark()
                for k, v in zip(it[0::2], it[1::2]):
                    self._setitem(st[-1], k, v)
            elif n == "MEMOIZE":
                self.memo[self.memo_next] = st[-1]; self.memo_next += 1
            elif n in ("BINPUT", "LONG_BINPUT", "PUT"):
                self.memo[int(arg)] = st[-1]
            elif n in ("BINGET", "LONG_BINGET", "GET"):
                st.append(self.memo[int(arg)])
            elif n == "GLOBAL":
                st.append(Sym("global", *arg.split(" ")))
            elif n == "STACK_GLOBAL":
                name = st.pop(); mod = st.pop(); st.append(Sym("global", mod, name))
            elif n == "REDUCE":
                args = st.pop(); fn = st.pop(); st.append(Sym("reduce", fn, args))
            elif n == "NEWOBJ":
                args = st.pop(); cls = st.pop(); st.append(Sym("newobj", cls, args))
            elif n == "BUILD":
                state = st.pop()
                obj = st[-1]
                if isinstance(obj, Sym):
                    obj.state = state
                    if self._is_wrapper(obj):
                        st[-1] = self._read_array(obj)
                else:
                    raise ValueError("BUILD on non-symbolic object")
            else:
                raise ValueError(f"opcode {n} not supported by the static reader")
        raise ValueError("pickle stream ended without STOP")

    @staticmethod
    def _setitem(d, k, v):
        if isinstance(d, dict):
            d[k] = v
        else:
            raise ValueError("SETITEM on non-dict")

    @staticmethod
    def _is_wrapper(sym):
        if sym.kind != "newobj":
            return False
        cls = sym.parts[0]
        return isinstance(cls, Sym) and cls.parts[-1] == "NumpyArrayWrapper"

    @staticmethod
    def _dtype_str(d):
        # numpy dtype pickles as reduce(global numpy dtype, ('f8', False, True)) + state
        if isinstance(d, Sym) and d.kind == "reduce":
            code = d.parts[1][0]
            state = d.state
            order = state[1] if state else "|"
            return code, order
        raise ValueError(f"unrecognised dtype record {d!r}")

    def _read_array(self, wrap):
        meta = wrap.state
        shape = tuple(int(x) for x in meta["shape"])
        code, order = self._dtype_str(meta["dtype"])
        if meta.get("order", "C") != "C":
            raise ValueError("only C-order arrays expected")
        if "numpy_array_alignment_bytes" in meta and meta["numpy_array_alignment_bytes"]:
            pad = self.fh.read(1)[0]
            self.fh.read(pad)
        count = int(np.prod(shape)) if shape else 1
        if code.startswith("O"):
            # object array: joblib appends a nested plain pickle of the ndarray
            sub = _Machine(self.fh).run()
            return _object_array_items(sub, count)
        if code[0] == "U" and code[1:].isdigit():
            dt = np.dtype((">" if order == ">" else "<") + code)
        elif code in ("u1", "i1", "b1"):
            dt = np.dtype(code)
        elif code in ("i2", "i4", "i8", "u2", "u4", "u8", "f4", "f8"):
            dt = np.dtype((">" if order == ">" else "<") + code)
        else:
            raise ValueError(f"dtype {code!r} not expected in a utmos .jl file")
        raw = self.fh.read(count * dt.itemsize)
        if len(raw) != count * dt.itemsize:
            raise ValueError("short read of array payload")
        return np.frombuffer(raw, dtype=dt).reshape(shape).copy()


def _object_array_items(sym, count):
    """An object ndarray pickles as reduce(_reconstruct, ...) with state
    (version, shape, dtype, is_fortran, list_of_items)."""
    if not (isinstance(sym, Sym) and sym.kind == "reduce" and isinstance(sym.state, tuple)):
        raise ValueError("unexpected object-array record")
    items = sym.state[-1]
    if not isinstance(items, list) or len(items) != count:
        raise ValueError("object array item count mismatch")
    if not all(isinstance(x, str) for x in items):
        raise ValueError("non-string item in object array")
    return np.array(items, dtype=str)


def read_jl(path):
    """Return {'GT': uint8 (n, ceil(S/8)), 'AF': float64 (n, 1), 'samples': str (S,), 'stats': dict}."""
    raw = open(path, "rb").read()
    try:
        raw = zlib.decompress(raw)
    except zlib.error:
        pass  # uncompressed joblib file
    top = _Machine(io.BytesIO(raw)).run()
    if not isinstance(top, dict):
        raise ValueError("top-level object is not a dict")
    return top


if __name__ == "__main__":
    import sys
    d = read_jl(sys.argv[1])
    for k, v in d.items():
        print(k, getattr(v, "shape", None), getattr(v, "dtype", None), v if not hasattr(v, "shape") else "")
