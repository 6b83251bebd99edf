"""Device-side input preparation for the NYU-Depth-v2 evaluation path (the reference's `dataloaders` package is CPU code on
removed SciPy / NumPy APIs and is not rebuilt; only the validation transform's arithmetic is kept, as an index map)."""
