#!/usr/bin/env python3
"""Cut the reference's host-only functions out of its two .cu files, whole and unmodified, so
that plain g++ can compile them (the files themselves need nvcc, cuRAND and Thrust).

    slice_reference.py <reference root> <output file>

TEST INFRASTRUCTURE.  The output is reference source text: it is written where the caller says
(the Makefile uses a temporary directory) and is never committed.  A function is taken only if
its body uses nothing beyond the C/C++ standard library; nothing is edited and no stand-in for
any CUDA name is supplied -- whatever would need one stays out.
"""
import re
import sys

# file -> [(function name, the reference lines it occupies, for the record)]
WANTED = {
    "imageAnalysisLib/imageAnalysis.cu": [
        ("random", "39-41"), ("equalizeHistogram", "155-176"), ("centroidsHaveChanged", "281-293"),
        ("arraySum", "682-688"),
    ],
    "imageAnalysisLib/imageMachineLearningLib/imageMachineLearning.cu": [
        ("#define NUM_CLASSES", "212"), ("computeConfusionMatrix", "214-227"), ("computeAccuracy", "229-236"),
        ("computePrecisionAndRecall", "239-255"), ("calcMode", "341-351"), ("calculateCellArea", "353-360"),
        ("calculateCellPerimeter", "362-366"), ("calcMinAndMaxOfRange", "379-391"),
    ],
}


def cut_function(text, name):
    """The definition `<type> name(...) { ... }` at file scope, by brace matching."""
    m = re.search(r"^[A-Za-z_][\w \*]*\b" + re.escape(name) + r"\s*\([^;{]*\)\s*\{", text, re.M)
    if not m:
        raise SystemExit(f"{name}: definition not found")
    depth, i = 0, m.end() - 1
    while True:
        c = text[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return text[m.start():i + 1]
        i += 1


def main():
    root, out = sys.argv[1], sys.argv[2]
    parts = []
    for rel, names in WANTED.items():
        text = open(f"{root}/{rel}").read()
        for name, lines in names:
            if name.startswith("#define"):
                m = re.search(r"^" + re.escape(name) + r".*$", text, re.M)
                if not m:
                    raise SystemExit(f"{name}: not found")
                parts.append(f"// {rel}:{lines}\n{m.group(0)}\n")
            else:
                parts.append(f"// {rel}:{lines}\n{cut_function(text, name)}\n")
    open(out, "w").write("\n".join(parts))


if __name__ == "__main__":
    main()
