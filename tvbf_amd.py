"""Import shim: the product package lives in the directory
``tvbingefriend-recommendation-service_amd/`` whose name is not a Python
identifier.  Importing this module registers it as
``tvbingefriend_recommendation_service_amd`` (sub-modules import normally
afterwards) and re-exports it.
"""

from __future__ import annotations

import importlib.util
import sys
from pathlib import Path

PACKAGE_NAME = "tvbingefriend_recommendation_service_amd"
PACKAGE_DIR = Path(__file__).resolve().parent / "tvbingefriend-recommendation-service_amd"


def load_package():
    if PACKAGE_NAME in sys.modules:
        return sys.modules[PACKAGE_NAME]
    spec = importlib.util.spec_from_file_location(
        PACKAGE_NAME, PACKAGE_DIR / "__init__.py", submodule_search_locations=[str(PACKAGE_DIR)]
    )
    module = importlib.util.module_from_spec(spec)
    sys.modules[PACKAGE_NAME] = module
    try:
        spec.loader.exec_module(module)
    except BaseException:
        sys.modules.pop(PACKAGE_NAME, None)
        raise
    return module


package = load_package()
