"""Re-export of synthetic/build.py (tests import the product-model builder from here)."""
import synthetic.build as _b

globals().update({k: v for k, v in vars(_b).items() if not k.startswith("__")})
