"""Re-export of synthetic/recipes.py (the oracle and the tests historically import the recipes from here)."""
import synthetic.recipes as _r

globals().update({k: v for k, v in vars(_r).items() if not k.startswith("__")})
