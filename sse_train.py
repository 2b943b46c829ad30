#!/usr/bin/env python
"""`python sse_train.py --flag=value ...` -- same command line as the reference's sse_train.py; runs the MI355X path."""
import sse_amd.sse_train as _cli

if __name__ == "__main__":
    _cli.main()
