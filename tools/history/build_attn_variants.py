"""Diagnostic builds of attention_ab.hip (only that unit is recompiled): python tools/build_attn_variants.py name=-Dflag[,-Dflag] ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import build
build.build_lib()
for arg in sys.argv[1:]:
    name, flags = arg.split("=", 1)
    print(build.build_variant_lib(name, [f for f in flags.split(",") if f], only=("attention_ab.hip", "attention.hip")), flush=True)
