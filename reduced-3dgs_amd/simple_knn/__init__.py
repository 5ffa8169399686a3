"""`simple_knn` -- nearest-neighbour operators of the reference's submodules/simple-knn, on the MI355X library."""
