"""Runs one attention case of tests/test_parity_configs_gpu.py in this process: python tools/attn_case_probe.py Lk masked mode"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_parity_configs_gpu as t  # noqa: E402
Lk, masked, mode = int(sys.argv[1]), sys.argv[2] == "1", int(sys.argv[3])
print("case", Lk, masked, mode, "err", t._attention_case(Lk, masked, mode), flush=True)
