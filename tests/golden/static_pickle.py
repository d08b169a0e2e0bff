"""Data-only reader for pickle streams that executes nothing from the file.

The stream is walked with ``pickletools.genops`` (a disassembler).  Containers and
scalars (dict, list, tuple, str, bytes, int, float, bool, None) are rebuilt directly.
A GLOBAL / STACK_GLOBAL opcode never imports anything: it becomes a ``Sym`` record that
just remembers the two names.  REDUCE / BUILD never call anything: they become ``Call``
records.  After the walk, ``materialise`` recognises exactly two shapes by *name* and
decodes them itself from the raw bytes they carry:

* ``numpy[._]core.multiarray._reconstruct`` + BUILD state  -> ``np.frombuffer`` array
  (little/big-endian fixed-width numeric dtypes only, C or Fortran order);
* ``numpy[._]core.multiarray.scalar(dtype, bytes)``       -> numpy scalar from bytes.

Anything else stays an inert ``Sym`` / ``Call`` record, so an unexpected payload can be
inspected but can never run.  This module is fixture-generation tooling only.
"""
from __future__ import annotations

import pickletools
from dataclasses import dataclass, field
from typing import Any

import numpy as np


@dataclass(frozen=True)
class Sym:
    module: str
    name: str


@dataclass
class Call:
    func: Any
    args: Any
    state: Any = None
    has_state: bool = False
    items: list = field(default_factory=list)


class _Mark:
    pass


_MARK = _Mark()


def parse(data: bytes) -> Any:
    """Return the inert object tree described by ``data``."""
    stack: list = []
    memo: dict = {}

    def pop_to_mark():
        out = []
        while True:
            v = stack.pop()
            if v is _MARK:
                break
            out.append(v)
        out.reverse()
        return out

    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        if n == "STOP":
            return stack.pop()
        if n == "MARK":
            stack.append(_MARK)
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n in ("BININT", "BININT1", "BININT2", "LONG1", "LONG4", "INT", "LONG"):
            stack.append(arg)
        elif n in ("BINFLOAT", "FLOAT"):
            stack.append(arg)
        elif n in ("SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8", "UNICODE"):
            stack.append(arg)
        elif n in ("SHORT_BINBYTES", "BINBYTES", "BINBYTES8", "BYTEARRAY8"):
            stack.append(bytes(arg))
        elif n in ("SHORT_BINSTRING", "BINSTRING", "STRING"):
            stack.append(arg)
        elif n == "MEMOIZE":
            memo[len(memo)] = stack[-1]
        elif n in ("BINPUT", "LONG_BINPUT", "PUT"):
            memo[arg] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET", "GET"):
            stack.append(memo[arg])
        elif n == "TUPLE":
            stack.append(tuple(pop_to_mark()))
        elif n == "TUPLE1":
            a = stack.pop()
            stack.append((a,))
        elif n == "TUPLE2":
            b = stack.pop()
            a = stack.pop()
            stack.append((a, b))
        elif n == "TUPLE3":
            c = stack.pop()
            b = stack.pop()
            a = stack.pop()
            stack.append((a, b, c))
        elif n == "LIST":
            stack.append(list(pop_to_mark()))
        elif n == "DICT":
            kv = pop_to_mark()
            stack.append(dict(zip(kv[0::2], kv[1::2])))
        elif n == "APPEND":
            v = stack.pop()
            _append(stack[-1], [v])
        elif n == "APPENDS":
            vs = pop_to_mark()
            _append(stack[-1], vs)
        elif n == "SETITEM":
            v = stack.pop()
            k = stack.pop()
            stack[-1][k] = v
        elif n == "SETITEMS":
            kv = pop_to_mark()
            d = stack[-1]
            for k, v in zip(kv[0::2], kv[1::2]):
                d[k] = v
        elif n == "GLOBAL":
            mod, name = arg.split(" ", 1)
            stack.append(Sym(mod, name))
        elif n == "STACK_GLOBAL":
            name = stack.pop()
            mod = stack.pop()
            stack.append(Sym(mod, name))
        elif n == "REDUCE":
            args = stack.pop()
            func = stack.pop()
            stack.append(Call(func, args))
        elif n == "NEWOBJ":
            args = stack.pop()
            cls = stack.pop()
            stack.append(Call(cls, args))
        elif n == "BUILD":
            state = stack.pop()
            obj = stack[-1]
            if isinstance(obj, Call):
                obj.state = state
                obj.has_state = True
            else:
                raise ValueError("BUILD on a non-symbolic object is not supported")
        elif n == "POP":
            stack.pop()
        elif n == "DUP":
            stack.append(stack[-1])
        else:
            raise ValueError(f"opcode {n} is outside the data-only subset")
    raise ValueError("pickle stream ended without STOP")


def _append(target, values):
    if isinstance(target, list):
        target.extend(values)
    elif isinstance(target, Call):
        target.items.extend(values)
    else:
        raise ValueError("APPEND on unsupported target")


_NP_CORE = ("numpy.core.multiarray", "numpy._core.multiarray")


def _dtype_from_call(c: Call) -> np.dtype:
    # numpy.dtype('f8', False, True) then BUILD (3, '<', None, None, None, -1, -1, 0)
    if not (isinstance(c, Call) and isinstance(c.func, Sym)
            and c.func.module == "numpy" and c.func.name == "dtype"):
        raise ValueError("not a numpy dtype record")
    code = c.args[0]
    if not (isinstance(code, str) and len(code) >= 2 and code[0] in "fiub"
            and code[1:].isdigit()):
        raise ValueError(f"dtype {code!r} outside the fixed-width numeric subset")
    order = "="
    if c.has_state and isinstance(c.state, tuple) and len(c.state) > 1:
        order = c.state[1]
    if order not in ("<", ">", "=", "|"):
        raise ValueError("bad byte order")
    return np.dtype(code).newbyteorder(order) if order in "<>" else np.dtype(code)


def materialise(node: Any, _seen=None) -> Any:
    """Turn recognised numpy records into arrays; recurse through containers."""
    if _seen is None:
        _seen = {}
    key = id(node)
    if key in _seen:
        return _seen[key]
    if isinstance(node, dict):
        out: Any = {}
        _seen[key] = out
        for k, v in node.items():
            out[materialise(k, _seen)] = materialise(v, _seen)
        return out
    if isinstance(node, list):
        out = []
        _seen[key] = out
        out.extend(materialise(v, _seen) for v in node)
        return out
    if isinstance(node, tuple):
        return tuple(materialise(v, _seen) for v in node)
    if isinstance(node, Call) and isinstance(node.func, Sym):
        f = node.func
        if f.module in _NP_CORE and f.name == "_reconstruct" and node.has_state:
            # state = (version, shape, dtype, is_fortran, rawdata)
            _ver, shape, dt, fortran, raw = node.state
            dtype = _dtype_from_call(dt)
            arr = np.frombuffer(raw, dtype=dtype).astype(dtype.newbyteorder("="))
            arr = arr.reshape(shape, order="F" if fortran else "C")
            _seen[key] = arr
            return arr
        if f.module in _NP_CORE and f.name == "scalar":
            dtype = _dtype_from_call(node.args[0])
            val = np.frombuffer(node.args[1], dtype=dtype)[0]
            _seen[key] = val
            return val
    return node


def load(path: str) -> Any:
    with open(path, "rb") as fh:
        return materialise(parse(fh.read()))
