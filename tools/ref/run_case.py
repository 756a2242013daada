#!/usr/bin/env python3
"""Runs the elprep command lines of ONE case directory (tools/ref/cases.py: commands()).  usage: run_case.py <elprep binary> <case dir>"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.ref import cases  # noqa: E402

elprep, w = sys.argv[1], sys.argv[2].rstrip("/")
case = json.load(open(os.path.join(w, "case.json")))
os.makedirs(os.path.join(w, "tmp"), exist_ok=True)
for argv in cases.commands(case, elprep, w):
    print("+", " ".join(argv), flush=True)
    subprocess.check_call(argv)
