"""Drop-in SAM image encoder (boundary b4 of SURVEY.md section 8)."""
