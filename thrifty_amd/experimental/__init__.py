"""Experimental detector variants (counterpart of reference thrifty/experimental/)."""
