"""Inert stand-ins for hydra.core (imported at module level by the reference's eval scripts; no arithmetic)."""
