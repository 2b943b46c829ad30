#!/usr/bin/env python
"""`python sse_demo.py --flag=value ...` -- same command line as the reference's sse_demo.py; runs the MI355X path."""
import sse_amd.sse_demo as _cli

if __name__ == "__main__":
    _cli.main()
