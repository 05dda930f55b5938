# Mirror package: the sub-modules found here replace the reference's; every other sub-module of this package keeps
# resolving to the reference checkout behind the mirror on sys.path (see xas_amd/_next.py).
__path__ = __import__('pkgutil').extend_path(__path__, __name__)
