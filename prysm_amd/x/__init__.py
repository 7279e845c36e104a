"""Experimental-module counterparts (prysm/x) that sit on the hot path."""
