"""Stand-in: the reference's URDF parser imports ``lxml.etree`` for ``get_urdf_string`` only (not used by the loader tests)."""
from . import etree  # noqa: F401
