"""Drop-in ISM proposal-vs-template scoring (boundary b3 of SURVEY.md section 8)."""
