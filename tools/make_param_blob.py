#!/usr/bin/env python3
"""Convert a ViennaRNA v2.0 parameter text file into the engine's int32 blob.

    python tools/make_param_blob.py /root/reference/rna_turner1999.par desirna_amd/data/turner1999.drnp

The shipped ``desirna_amd/data/turner1999.drnp`` was produced by exactly this command from the
reference's ``rna_turner1999.par`` (Turner-1999 nearest-neighbour parameters as distributed with
ViennaRNA; data, not code).  Only the 37 degC free energies are kept (see desirna_amd/params.py).
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from desirna_amd import params  # noqa: E402


def main():
    src, dst = sys.argv[1], sys.argv[2]
    blob = params.load_par_file(src)
    params.save_blob(blob, dst)
    print("wrote %s: %d int32 (%d bytes)" % (dst, blob.size, blob.size * 4))


if __name__ == "__main__":
    main()
