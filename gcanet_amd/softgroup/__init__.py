"""Drop-in namespace for the reference's ``softgroup`` package (only ``softgroup.ops``)."""
