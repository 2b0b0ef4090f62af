#!/usr/bin/env python3
"""Extract the first 400 training rows of the reference's `holes` data set into a
small CSV fixture (BASELINE config C1 input).

Run in the BUILD container only (reads /root/reference/data/holes.rda, which does
not travel to the GPU box).  The .rda is gzip + R "RDX3" XDR serialization; this
is a minimal reader for the node types that file contains (pairlist, generic
vector, real/int/string vectors, symbols, attributes).

Usage: python tests/golden/make_holes_fixture.py
Output: tests/golden/holes_train400.csv   (columns x,y,cov_x,cov_y,z)
"""
import gzip
import os
import struct
import sys

SRC = "/root/reference/data/holes.rda"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "holes_train400.csv")


class Reader:
    def __init__(self, b):
        self.b = b
        self.o = 0
        self.refs = []

    def i32(self):
        (v,) = struct.unpack_from(">i", self.b, self.o)
        self.o += 4
        return v

    def f64n(self, n):
        v = struct.unpack_from(">%dd" % n, self.b, self.o)
        self.o += 8 * n
        return list(v)

    def raw(self, n):
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def item(self):
        flags = self.i32()
        ty = flags & 0xFF
        has_attr = bool(flags & (1 << 9))
        has_tag = bool(flags & (1 << 10))
        if ty == 254:      # NILVALUE_SXP
            return None
        if ty == 253:      # GLOBALENV
            return "<globalenv>"
        if ty == 255:      # REFSXP
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if ty == 1:        # SYMSXP
            name = self.item()
            self.refs.append(name)
            return name
        if ty == 2:        # LISTSXP (pairlist)
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                flags = self.i32()
                ty2 = flags & 0xFF
                if ty2 == 254:
                    break
                if ty2 != 2:
                    raise ValueError("unexpected cdr type %d" % ty2)
                has_attr = bool(flags & (1 << 9))
                has_tag = bool(flags & (1 << 10))
            return out
        if ty == 9:        # CHARSXP
            n = self.i32()
            return None if n == -1 else self.raw(n).decode("utf-8", "replace")
        if ty == 10 or ty == 13:   # LGLSXP / INTSXP
            n = self.i32()
            v = [self.i32() for _ in range(n)]
            res = {"int": v}
        elif ty == 14:     # REALSXP
            n = self.i32()
            res = {"real": self.f64n(n)}
        elif ty == 16:     # STRSXP
            n = self.i32()
            res = {"str": [self.item() for _ in range(n)]}
        elif ty == 19:     # VECSXP
            n = self.i32()
            res = {"list": [self.item() for _ in range(n)]}
        else:
            raise ValueError("unsupported SEXP type %d at %d" % (ty, self.o))
        if has_attr:
            res["attr"] = dict(self.item())
        return res


def main():
    if not os.path.exists(SRC):
        sys.exit("reference data not present; fixture is already committed")
    b = gzip.decompress(open(SRC, "rb").read())
    assert b[:5] == b"RDX3\n" and b[5:7] == b"X\n"
    r = Reader(b)
    r.o = 7
    r.i32(); r.i32(); r.i32()          # format version, writer R version, min R version
    n = r.i32(); r.raw(n)              # native encoding
    top = r.item()                     # pairlist of (name, value)
    holes = dict(top)["holes"]
    names = holes["attr"]["names"]["str"]
    training = holes["list"][names.index("training")]
    cols = training["attr"]["names"]["str"]
    data = {c: training["list"][i]["real"] for i, c in enumerate(cols)}
    nrow = len(data[cols[0]])
    test = holes["list"][names.index("test")]
    print("training rows", nrow, "cols", cols, "test rows", len(test["list"][0]["real"]))
    want = ["x", "y", "cov_x", "cov_y", "z"]
    with open(OUT, "w") as f:
        f.write(",".join(want) + "\n")
        for i in range(400):
            f.write(",".join(repr(float(data[c][i])) for c in want) + "\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
