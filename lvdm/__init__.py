"""Drop-in import path of the reference (`lvdm...`): only the modules of the encode/decode path and its frozen 2-D constraint decoder exist here."""
