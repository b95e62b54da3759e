from .lib import HipError, dtype_code, get_lib, lib_path  # noqa: F401
