"""Import-path alias of the reference's Python package: ``import deltaRice.h5`` keeps working when the
MI355X codec replaces the CPU filter.  Everything lives in :mod:`deltarice_amd`."""
