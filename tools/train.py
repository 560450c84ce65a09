#!/usr/bin/env python3
"""Entry point for the runner (package directory name has hyphens, so it is imported via importlib)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd.runner").main()
