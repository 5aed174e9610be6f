"""Front-ends over the engine surface (the counterpart of the reference's ``app/``): a terminal chat loop, the wire-API
demo, a streaming web chat.  Nothing here touches the hot path -- they call ``prefill / append / speculative_decoding /
generate / generate_stream`` exactly as the reference's scripts do."""
