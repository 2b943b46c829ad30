#!/usr/bin/env python
"""`python sse_index.py --flag=value ...` -- same command line as the reference's sse_index.py; runs the MI355X path."""
import sse_amd.sse_index as _cli

if __name__ == "__main__":
    _cli.main()
