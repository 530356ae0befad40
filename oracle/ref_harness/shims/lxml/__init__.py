from . import etree  # noqa
