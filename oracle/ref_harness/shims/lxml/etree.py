import xml.etree.ElementTree as _ET
class _Elem(_ET.Element):
    def getchildren(self):
        return list(self)
class XMLParser:
    def __init__(self, *a, **k):
        pass
def parse(source, parser=None):
    tb = _ET.TreeBuilder(element_factory=_Elem)
    p = _ET.XMLParser(target=tb)
    data = source.read() if hasattr(source, "read") else open(source, "rb").read()
    p.feed(data)
    root = p.close()
    return _ET.ElementTree(root)
ElementTree = _ET.ElementTree
Element = _Elem
SubElement = _ET.SubElement
