"""Reader for Fortran-90 array-constructor *data* (``name(1:50, 3) = (/ ... /)``).

This is how RRTMG_LW ships every table: the Planck integrals and reference atmosphere in
``rrtmg_lw_setcoef.f90``, the cloud coefficients in ``rrtmg_lw_init.f90`` and - in the data-statement
distribution of the absorption coefficients - ``rrtmg_lw_k_g.f90``.  Only assignments of literal
constants are understood; executable code is skipped.  Nothing here evaluates Fortran.
"""
from __future__ import annotations

import re

import numpy as np

_ASSIGN = re.compile(r"^\s*(\w+)\s*(?:\(([^=]*?)\))?\s*=\s*\(/(.*)/\)\s*$", re.S)
_SCALAR = re.compile(r"^\s*(\w+)\s*=\s*([-+]?[0-9.]+(?:[eEdD][-+]?[0-9]+)?)(?:_\w+)?\s*$")
_NUM = re.compile(r"([-+]?(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eEdD][-+]?[0-9]+)?)(?:_\w+)?")


def _statements(text):
    """Yield logical statements: comments stripped, ``&`` continuations joined."""
    cur = ""
    for raw in text.splitlines():
        line = raw.split("!", 1)[0].rstrip()
        if not line.strip():
            continue
        s = line.strip()
        if s.startswith("&"):
            s = s[1:].lstrip()
        if s.endswith("&"):
            cur += s[:-1] + " "
            continue
        cur += s
        yield cur
        cur = ""
    if cur.strip():
        yield cur


def _parse_values(body):
    vals = []
    for tok in body.split(","):
        tok = tok.strip()
        if not tok:
            continue
        m = _NUM.fullmatch(tok)
        if not m:
            raise ValueError(f"not a literal constant: {tok!r}")
        vals.append(float(m.group(1).replace("d", "e").replace("D", "e")))
    return vals


def parse_f90_data(text, shapes, scalars=(), routine=None):
    """Collect constant assignments to the arrays named in ``shapes``.

    shapes : name -> list of (lo, hi) Fortran bounds per dimension
    scalars: names of scalar variables to pick up (``abscld1 = 0.0602410_rb``)
    routine: if given, only statements between ``subroutine <routine>`` and its ``end subroutine``
    Returns (arrays, scalar_values); arrays have the Fortran shape, NaN where never assigned.
    """
    arrays = {n: np.full([hi - lo + 1 for lo, hi in b], np.nan) for n, b in shapes.items()}
    found = {}
    active = routine is None
    for st in _statements(text):
        low = st.lower()
        if routine is not None:
            if re.match(rf"^\s*subroutine\s+{routine}\b", low):
                active = True
                continue
            if active and re.match(r"^\s*end\s+subroutine", low):
                active = False
                continue
        if not active:
            continue
        m = _SCALAR.match(low)
        if m and m.group(1) in scalars:
            found[m.group(1)] = float(m.group(2).replace("d", "e"))
            continue
        m = _ASSIGN.match(low)
        if not m or m.group(1) not in shapes:
            continue
        name, sl, body = m.group(1), m.group(2), m.group(3)
        bounds = shapes[name]
        vals = _parse_values(body)
        if sl is None:
            sl = ",".join(":" for _ in bounds)
        parts = [p.strip() for p in sl.split(",")]
        if len(parts) != len(bounds):
            raise ValueError(f"{name}: rank mismatch in '{st[:60]}'")
        index = []
        count = 1
        for p, (lo, hi) in zip(parts, bounds):
            if p == ":":
                a, b = lo, hi
            elif ":" in p:
                a, b = p.split(":")
                a = int(a) if a.strip() else lo
                b = int(b) if b.strip() else hi
            else:
                a = b = int(p)
            if a < lo or b > hi:
                raise ValueError(f"{name}: slice {p} outside bounds {lo}:{hi}")
            index.append(slice(a - lo, b - lo + 1))
            count *= b - a + 1
        if count != len(vals):
            raise ValueError(f"{name}({sl}): {len(vals)} values for {count} elements")
        view = arrays[name][tuple(index)]
        view[...] = np.asarray(vals).reshape(view.shape, order="F")
    return arrays, found
