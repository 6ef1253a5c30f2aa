#!/usr/bin/env python3
"""code object -> comma-separated bytes, to be #included inside a `static const unsigned char x[] = { ... };`."""
import sys
data = open(sys.argv[1], "rb").read()
with open(sys.argv[2], "w") as f:
    for i in range(0, len(data), 32):
        f.write(",".join(str(b) for b in data[i:i + 32]) + ",\n")
