# extend the package over the reference checkout that follows on sys.path
from pkgutil import extend_path
__path__ = extend_path(__path__, __name__)
