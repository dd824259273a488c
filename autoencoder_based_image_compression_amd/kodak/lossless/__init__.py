"""kodak/lossless: part of the MI355X build of the compression inference path."""
