"""Extract the expected main-trace tables of the reference's own unit tests into a JSON fixture.

Source of the vectors (run in the build container only; the fixture is what travels):
  /root/reference/src/cairo/execution_trace.rs:661-768   test_build_main_trace_simple_program   (3 steps)
  /root/reference/src/cairo/execution_trace.rs:771-1161  test_build_main_trace_call_func_program (7 steps)
Each `expected_trace` there is a list of 34 columns of FE::zero()/FE::one()/FE::from(<int>)/FE::from_hex_unchecked("..").
The program words are not in the reference tree (only the .cairo sources are); the test derives them from the tables
themselves (instruction column 23 at pc = column 19, immediates = op1 when op1_addr = pc + 1).

Usage: python tests/golden/make_main_trace_vectors.py  ->  tests/golden/main_trace_tables.json
"""
import json
import os
import re

SRC = "/root/reference/src/cairo/execution_trace.rs"
HERE = os.path.dirname(os.path.abspath(__file__))

TOKEN = re.compile(r'FE::zero\(\)|FE::one\(\)|FE::from\(\s*(0x[0-9a-fA-F]+|\d+)\s*\)|FE::from_hex_unchecked\(\s*"([0-9a-fA-F]+)",?\s*\)')


def parse_table(text):
    cols = []
    for chunk in re.split(r"//\s*col \d+[^\n]*\n", text)[1:]:
        body = chunk[chunk.index("vec!["):]
        depth, end = 0, None
        for i, ch in enumerate(body):
            if ch == "[":
                depth += 1
            elif ch == "]":
                depth -= 1
                if depth == 0:
                    end = i
                    break
        vals = []
        for m in TOKEN.finditer(body[:end]):
            t = m.group(0)
            if t.startswith("FE::zero"):
                vals.append(0)
            elif t.startswith("FE::one"):
                vals.append(1)
            elif m.group(1):
                vals.append(int(m.group(1), 0))
            else:
                vals.append(int(m.group(2), 16))
        cols.append([hex(v) for v in vals])
    return cols


def main():
    src = open(SRC).read()
    out = {}
    for name in ("simple_program", "call_func_program"):
        start = src.index(f"fn test_build_main_trace_{name}()")
        end = src.index("assert_eq!(execution_trace.cols(), expected_trace.cols());", start)
        body = src[src.index("let expected_trace", start):end]
        cols = parse_table(body)
        assert len(cols) == 34 and len({len(c) for c in cols}) == 1, (name, len(cols))
        out[name] = {"columns": cols}
    with open(os.path.join(HERE, "main_trace_tables.json"), "w") as f:
        json.dump(out, f, indent=0)
    print({k: (len(v["columns"]), len(v["columns"][0])) for k, v in out.items()})


if __name__ == "__main__":
    main()
