"""Importable alias of the package directory `bystro-vcf_amd/` (a hyphen cannot be imported)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bystro-vcf_amd")
_spec = importlib.util.spec_from_file_location("bystro_vcf_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bystro_vcf_amd"] = _mod
_spec.loader.exec_module(_mod)
