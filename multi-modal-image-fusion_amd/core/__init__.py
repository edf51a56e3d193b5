"""Drop-in mirror of the reference's `core` package (model / block / fusion / loss) backed by the
MI355X HIP engine in ../mmif.  Same module paths, class names, constructor and forward signatures
and state_dict keys as chenzpstar/Multi-Modal-Image-Fusion core/*.py."""
