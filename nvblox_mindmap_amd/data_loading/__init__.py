from .vertex_sampling import VertexSamplingMethod, sample_to_n_vertices  # noqa: F401
