"""Host-side model definition: configuration, parameter schema, checkpoint adapters."""
