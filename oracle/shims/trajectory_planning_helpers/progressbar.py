def progressbar(i, i_total, prefix="", suffix="", decimals=1, length=50):
    """Console progress bar in tph; silent in the oracle shim."""
    return None
